#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python tools/tpt_check.py quick > gpurun_out/tpt_quick.log 2>&1
rc=$?; echo "rc=$rc"; grep -c "pixels ==" gpurun_out/tpt_quick.log; tail -1 gpurun_out/tpt_quick.log
CLWRAP_TPT_CLOCK=1 timeout -k 10 600 python tools/tpt_check.py time ref800,hd15,c3s 0,32,48,64 2>&1 | cut -c1-520 > gpurun_out/tpt_time3.log
cat gpurun_out/tpt_time3.log
