#!/bin/bash
# round 3: VIS kernels (visibility classes + compacted shadow rays) -- parity, then A/B against the batch kernels
mkdir -p gpurun_out; rm -f gpurun_out/parity_report.jsonl gpurun_out/vis_ab_r03b.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -q -m gpu -p no:cacheprovider -x -k "visibility or c3_c4 or golden or raypng or two_kernel or ragged or strips" > gpurun_out/pytest_r03b_new.log 2>&1
echo "new tests rc=$?"; tail -15 gpurun_out/pytest_r03b_new.log
for c in c2 ref800; do for s in 0 1; do for v in 256 2048 0 256 0; do
  timeout -k 10 120 python3 tools/run_config.py $c --frames 60 --strict $s --variant $v | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config'],'strict',d['strict'],'variant',d['variant'],'kernel_ms',d['kernel_ms'],'traced',d['counters']['shadow_rays_traced'],'of',d['counters']['shadow_rays'],'classified',d['counters'].get('lights_classified'),'lane_util',d['lane_util'])" | tee -a gpurun_out/vis_ab_r03b.log
done; done; done
timeout -k 10 900 python -m pytest tests -q -m gpu -p no:cacheprovider -rfE > gpurun_out/pytest_r03b.log 2>&1
echo "suite rc=$?"; tail -15 gpurun_out/pytest_r03b.log
