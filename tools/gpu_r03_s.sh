#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_functions.py -q -m gpu -p no:cacheprovider -x -k "grid or cloud or many_spheres or nan_rays or full_size_c4 or c3_c4 or scene or tail_with_many" > gpurun_out/pytest_r03s.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/pytest_r03s.log | cut -c1-300
for p in 1 0 1 0; do echo -n "pair=$p: "; CLWRAP_GRID_PAIR=$p python3 tools/run_config.py c4 --frames 30 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernel_ms'], d['Mrays_s'])"; done
for d in 1.2 1.6 2.0 2.4; do echo -n "density $d: "; CLWRAP_GRID_DENSITY=$d python3 tools/run_config.py c4 --frames 30 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernel_ms'], d['Mrays_s'])"; done
