#!/bin/bash
# round 2, GPU call B: suite again (after the bar fixes), phase stamps, rocprofv3 evidence for C2 / C3 / C4
mkdir -p gpurun_out; rm -f gpurun_out/parity_report.jsonl gpurun_out/stamps_r02b.log
python -m pytest tests -q -m gpu -p no:cacheprovider -rfE > gpurun_out/pytest_r02b.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/pytest_r02b.log
tail -15 gpurun_out/pytest_r02b.log
for a in "c2" "c2 --strict" "ref800" "c3"; do timeout -k 10 200 python tools/stamp_phases.py $a >> gpurun_out/stamps_r02b.log 2>&1; done; cat gpurun_out/stamps_r02b.log
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/bench_r02b.json 2> gpurun_out/bench_r02b.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/bench_r02b.json')); print({k: d[k] for k in ('value','ms_per_step','trace_kernel_ms','strict','moving_camera','frames_per_s_with_readback')})"
timeout -k 10 500 bash tools/profile_round.sh r02a && timeout -k 10 300 bash tools/profile_config.sh r02a c3 6 && timeout -k 10 300 bash tools/profile_config.sh r02a c4 8
