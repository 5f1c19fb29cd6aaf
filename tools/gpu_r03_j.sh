#!/bin/bash
# full GPU suite + the driver-style bench line (re-entry check of the restored checkpoint)
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -q -m gpu -p no:cacheprovider -x > gpurun_out/pytest_r03j.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/pytest_r03j.log | cut -c1-250
[ $rc -eq 0 ] && timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r03j.json 2> gpurun_out/bench_r03j.err
echo "bench rc=$?"; cut -c1-400 gpurun_out/bench_r03j.json
