#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/parity_report.jsonl gpurun_out/stamps_r02c.log
python -m pytest tests -q -m gpu -p no:cacheprovider -rfE -x > gpurun_out/pytest_r02c.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/pytest_r02c.log
tail -12 gpurun_out/pytest_r02c.log
for a in "c2" "c2 --strict" "c3"; do timeout -k 10 200 python tools/stamp_phases.py $a >> gpurun_out/stamps_r02c.log 2>&1; done; cat gpurun_out/stamps_r02c.log
bash tools/ab_bench.sh "" _w4 _w6 2>&1 | tee gpurun_out/ab_r02c.log
for c in c3 c4 ref800; do python tools/run_config.py $c --frames 10 | cut -c1-200; done 2>&1 | tee gpurun_out/cfg_r02c.log
