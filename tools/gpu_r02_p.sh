#!/bin/bash
# final evidence of the round: suite, bench, rocprofv3 kernel stats + PMC passes for C2 (bench command), C3, C4, reference config
mkdir -p gpurun_out; rm -f gpurun_out/parity_report.jsonl gpurun_out/stamps_r02.log
python -m pytest tests -q -m gpu -p no:cacheprovider -rfE > gpurun_out/pytest_r02p.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/pytest_r02p.log; tail -5 gpurun_out/pytest_r02p.log
timeout -k 10 400 python bench.py --steps 300 --warmup 30 > gpurun_out/bench_r02p.json 2> gpurun_out/bench_r02p.err; echo "bench rc=$?"
for a in "c2" "c2 --strict" "ref800" "c3" "c4"; do timeout -k 10 200 python tools/stamp_phases.py $a >> gpurun_out/stamps_r02.log 2>&1; done; cat gpurun_out/stamps_r02.log
for c in c2 c3 c4 ref800 c5strip; do python3 tools/run_config.py $c --frames 30 | tee -a gpurun_out/configs_r02.jsonl | cut -c1-170; python3 tools/run_config.py $c --frames 30 --strict 1 | tee -a gpurun_out/configs_r02.jsonl | cut -c1-170; done
timeout -k 10 500 bash tools/profile_round.sh r02 && timeout -k 10 300 bash tools/profile_config.sh r02 c3 6 && timeout -k 10 300 bash tools/profile_config.sh r02 c4 10
