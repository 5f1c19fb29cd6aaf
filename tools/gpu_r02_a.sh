#!/bin/bash
# round 2, GPU call A: the whole -m gpu suite (no -x: every failure is wanted), the bench line, the phase stamps
mkdir -p gpurun_out
python -m pytest tests -q -m gpu -p no:cacheprovider -rfE --durations=15 > gpurun_out/pytest_r02a.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/pytest_r02a.log
tail -60 gpurun_out/pytest_r02a.log
timeout -k 10 400 python bench.py > gpurun_out/bench_r02a.json 2> gpurun_out/bench_r02a.err; echo "bench rc=$?"; cat gpurun_out/bench_r02a.json | cut -c1-1500
for a in "c2" "c2 --strict" "ref800" "c3"; do timeout -k 10 200 python tools/stamp_phases.py $a >> gpurun_out/stamps_r02a.log 2>&1; done; cat gpurun_out/stamps_r02a.log
