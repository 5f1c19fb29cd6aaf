#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/parity_report.jsonl
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -q -m gpu -p no:cacheprovider -x -k "visibility or golden or raypng or two_kernel or ragged or strips" 2>&1 | tail -5
for c in c2 ref800; do for s in 0 1; do for v in 256 0 256 0; do timeout -k 10 120 python3 tools/run_config.py $c --frames 60 --strict $s --variant $v | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['counters']; print(d['config'],'strict',d['strict'],'variant',d['variant'],d['kernel_ms'],'traced',c['shadow_rays_traced'],'of',c['shadow_rays'],'classified',c['lights_classified'])"; done; done; done
for v in 0 256; do timeout -k 10 200 python tools/stamp_phases.py c2 --variant $v; done
