#!/bin/bash
cd "$(dirname "$0")/.."
for round in 1 2; do
for p in 1 0; do
  echo -n "persist=$p:"
  for c in c2 c3 c4 ref800 c5strip; do
    CLWRAP_PERSIST=$p python3 tools/run_config.py $c --frames 30 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(' ', d['config'], d['kernel_ms'], end='')"
  done
  echo
done
echo -n "prev     :"; for c in c2 c3 c4 ref800 c5strip; do CLWRAP_LIB=$PWD/example_gui_opencl_raytracer_amd/libopencl_wrap_hip_prev.so python3 tools/run_config.py $c --frames 30 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(' ', d['config'], d['kernel_ms'], end='')"; done; echo
done
for p in 1 0; do echo -n "bench persist=$p: "; CLWRAP_PERSIST=$p python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['trace_kernel_ms'], 'strict', d['strict']['trace_kernel_ms'])"; done
