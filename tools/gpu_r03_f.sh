#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/parity_report.jsonl
timeout -k 10 1100 python -m pytest tests -q -m gpu -p no:cacheprovider -rfE -x > gpurun_out/pytest_r03f.log 2>&1
echo "suite rc=$?"; tail -12 gpurun_out/pytest_r03f.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r03f.json 2> gpurun_out/bench_r03f.err; echo "bench rc=$?"; cut -c1-1500 gpurun_out/bench_r03f.json; tail -3 gpurun_out/bench_r03f.err
for c in c2 ref800 c3 c4; do for s in 0 1; do timeout -k 10 200 python3 tools/run_config.py $c --frames 30 --strict $s | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['counters']; print(d['config'],'strict',d['strict'],d['kernel_ms'],'Mrays/s',d['Mrays_s'],'traced',c['shadow_rays_traced'],'of',c['shadow_rays'],'lane_util',d['lane_util'])"; done; done
