#!/bin/bash
# round 3, first GPU pass: new parity tests, visibility classes A/B (variant 256 = classes off), then the whole suite
mkdir -p gpurun_out; rm -f gpurun_out/parity_report.jsonl
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -x -k "visibility or c3_c4" > gpurun_out/pytest_r03a_new.log 2>&1
echo "new tests rc=$?"; tail -15 gpurun_out/pytest_r03a_new.log
for c in c2 ref800; do for s in 0 1; do for v in 256 0 256 0; do
  timeout -k 10 120 python3 tools/run_config.py $c --frames 60 --strict $s --variant $v | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config'],'strict',d['strict'],'variant',d['variant'],'kernel_ms',d['kernel_ms'],'traced',d['counters']['shadow_rays_traced'],'of',d['counters']['shadow_rays'],'classified',d['counters'].get('lights_classified'))" | tee -a gpurun_out/vis_ab_r03a.log
done; done; done
timeout -k 10 900 python -m pytest tests -q -m gpu -p no:cacheprovider -rfE > gpurun_out/pytest_r03a.log 2>&1
echo "suite rc=$?"; tail -15 gpurun_out/pytest_r03a.log
