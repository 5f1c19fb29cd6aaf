#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/tpt_c3.log
timeout -k 10 300 python tools/tpt_check.py quick > gpurun_out/tpt_quick.log 2>&1
rc=$?; echo "rc=$rc"; grep -c "pixels ==" gpurun_out/tpt_quick.log; tail -1 gpurun_out/tpt_quick.log
timeout -k 10 600 python tools/tpt_check.py time ref800,hd15,c3s 0,48 2>&1 | cut -c1-130 >> gpurun_out/tpt_c3.log
CLWRAP_TPT_CLOCK=1 timeout -k 10 600 python tools/tpt_check.py time ref800 48 2>&1 | cut -c1-420 >> gpurun_out/tpt_c3.log
cat gpurun_out/tpt_c3.log
