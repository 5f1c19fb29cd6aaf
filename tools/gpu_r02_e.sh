#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/parity_report.jsonl
python -m pytest tests -q -m gpu -p no:cacheprovider -rfE -x > gpurun_out/pytest_r02e.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/pytest_r02e.log
tail -8 gpurun_out/pytest_r02e.log
bash tools/ab_cfg2.sh "c2 c3 c4 ref800" "" _nochunk 2>&1 | tee gpurun_out/ab_r02e.log
bash tools/ab_bench.sh "" _nochunk 2>&1 | tee -a gpurun_out/ab_r02e.log
