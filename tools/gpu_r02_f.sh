#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/parity_report.jsonl
python -m pytest tests -q -m gpu -p no:cacheprovider -rfE > gpurun_out/pytest_r02f.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/pytest_r02f.log
tail -8 gpurun_out/pytest_r02f.log
for r in 1 2; do
for v in 0 128; do
  echo -n "variant $v fast: "; CLWRAP_VARIANT=$v python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['trace_kernel_ms'], 'strict', d['strict']['trace_kernel_ms'])"
done; done 2>&1 | tee gpurun_out/ab_r02f.log
for v in 0 128; do CLWRAP_VARIANT=$v python3 tools/run_config.py c3 --frames 20 | cut -c1-160; CLWRAP_VARIANT=$v python3 tools/run_config.py ref800 --frames 20 | cut -c1-160; done 2>&1 | tee -a gpurun_out/ab_r02f.log
