#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/libm_divergence.jsonl
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -v -m gpu -p no:cacheprovider -k "full_size_c2 or full_size_c3 or full_size_c4 or deep_refraction" > gpurun_out/pytest_r03i.log 2>&1
echo "rc=$?"; tail -12 gpurun_out/pytest_r03i.log | cut -c1-250; cat gpurun_out/libm_divergence.jsonl
