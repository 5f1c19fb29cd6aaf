#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -v -m gpu -p no:cacheprovider -x -k "pooled or large_frame" > gpurun_out/pytest_r03h.log 2>&1
echo "rc=$?"; tail -15 gpurun_out/pytest_r03h.log | cut -c1-220
