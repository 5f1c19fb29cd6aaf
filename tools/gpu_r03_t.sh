#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python tools/tpt_check.py quick > gpurun_out/tpt_quick.log 2>&1
rc=$?; echo "rc=$rc"; grep -c "pixels ==" gpurun_out/tpt_quick.log; tail -1 gpurun_out/tpt_quick.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -x -k "tree_parallel or deep_refraction or depths" > gpurun_out/pytest_r03t.log 2>&1
echo "pytest rc=$?"; tail -2 gpurun_out/pytest_r03t.log | cut -c1-250
timeout -k 10 600 python tools/tpt_check.py time ref800,hd15,c3s 48 2>&1 | cut -c1-100
CLWRAP_TPT_CLOCK=1 timeout -k 10 600 python tools/tpt_check.py time ref800 48 2>&1 | cut -c100-420
