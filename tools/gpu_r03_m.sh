#!/bin/bash
mkdir -p gpurun_out
./tools/ubench/xcc_probe > gpurun_out/xcc_probe.log 2>&1; cat gpurun_out/xcc_probe.log | head -8
timeout -k 10 400 python tools/tpt_check.py quick > gpurun_out/tpt_quick.log 2>&1
rc=$?; echo "rc=$rc"; grep -c "pixels ==" gpurun_out/tpt_quick.log; grep "!=" gpurun_out/tpt_quick.log | cut -c1-260; tail -3 gpurun_out/tpt_quick.log | cut -c1-250
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python tools/tpt_check.py time ref800,hd15,c3s 0,16,24,32,48,64 > gpurun_out/tpt_time.log 2>&1
echo "time rc=$?"; cat gpurun_out/tpt_time.log | cut -c1-250
