#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -x -k "deep_refraction or large_frame or full_size_c3 or c3_c4 or glass_field or many_spheres or maximum_one_byte or tree_parallel" > gpurun_out/pytest_r03t.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/pytest_r03t.log | cut -c1-250
for i in 1 2 3; do python3 tools/run_config.py c3 --frames 8 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3', d['kernel_ms'], d['Mrays_s'])"; done
python3 tools/run_config.py c3 --frames 6 --strict 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3 strict', d['kernel_ms'], d['Mrays_s'])"
