#!/bin/bash
mkdir -p gpurun_out
CLWRAP_OCC_TILES_PER_DEPTH=1 timeout -k 10 300 python tools/tpt_check.py quick > gpurun_out/tpt_quick_occ.log 2>&1
rc=$?; echo "occ rc=$rc"; grep -c "pixels ==" gpurun_out/tpt_quick_occ.log; grep "!=" gpurun_out/tpt_quick_occ.log | head -5 | cut -c1-250; tail -2 gpurun_out/tpt_quick_occ.log | cut -c1-250
timeout -k 10 300 python tools/tpt_check.py quick > gpurun_out/tpt_quick.log 2>&1
rc=$?; echo "rc=$rc"; grep -c "pixels ==" gpurun_out/tpt_quick.log; tail -1 gpurun_out/tpt_quick.log
timeout -k 10 600 python tools/tpt_check.py time c3 0,16,32,48,56 2>&1 | cut -c1-150 > gpurun_out/tpt_c3.log
cat gpurun_out/tpt_c3.log
