#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/parity_report.jsonl gpurun_out/vis_ab_r03d.log gpurun_out/stamps_r03d.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -q -m gpu -p no:cacheprovider -x -k "visibility or golden or raypng or two_kernel or ragged or strips" > gpurun_out/pytest_r03d_new.log 2>&1
echo "new tests rc=$?"; tail -8 gpurun_out/pytest_r03d_new.log
for c in c2 ref800; do for s in 0 1; do for v in 256 0 256 0; do
  timeout -k 10 120 python3 tools/run_config.py $c --frames 60 --strict $s --variant $v | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config'],'strict',d['strict'],'variant',d['variant'],'kernel_ms',d['kernel_ms'],'traced',d['counters']['shadow_rays_traced'],'of',d['counters']['shadow_rays'],'classified',d['counters'].get('lights_classified'))" | tee -a gpurun_out/vis_ab_r03d.log
done; done; done
for v in 0 256; do timeout -k 10 200 python tools/stamp_phases.py c2 --variant $v >> gpurun_out/stamps_r03d.log 2>&1; timeout -k 10 200 python tools/stamp_phases.py c2 --strict --variant $v >> gpurun_out/stamps_r03d.log 2>&1; done
cat gpurun_out/stamps_r03d.log
