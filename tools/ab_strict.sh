#!/bin/bash
cd "$(dirname "$0")/.."
for round in 1 2; do
for lib in "$@"; do
  if [ -n "$lib" ]; then export CLWRAP_LIB=$PWD/example_gui_opencl_raytracer_amd/libopencl_wrap_hip$lib.so; else unset CLWRAP_LIB; fi
  echo -n "lib='$lib': "
  python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['trace_kernel_ms'], 'strict', d['strict']['trace_kernel_ms'])"
done
done
