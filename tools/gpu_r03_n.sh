#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python tools/tpt_check.py quick > gpurun_out/tpt_quick.log 2>&1
rc=$?; echo "rc=$rc"; grep -c "pixels ==" gpurun_out/tpt_quick.log; tail -1 gpurun_out/tpt_quick.log
timeout -k 10 600 python tools/tpt_check.py time ref800,hd15,c3s 0,24,48 > gpurun_out/tpt_time.log 2>&1
echo "time rc=$?"; cat gpurun_out/tpt_time.log | cut -c1-400
