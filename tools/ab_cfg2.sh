#!/bin/bash
# A/B of tagged library builds on several configurations (kernel ms); usage: tools/ab_cfg2.sh "cfgs" lib1 lib2 ...
cd "$(dirname "$0")/.."
cfgs=$1; shift
for round in 1 2; do
for lib in "$@"; do
  if [ -n "$lib" ]; then export CLWRAP_LIB=$PWD/example_gui_opencl_raytracer_amd/libopencl_wrap_hip$lib.so; else unset CLWRAP_LIB; fi
  echo -n "lib='$lib':"
  for c in $cfgs; do
    python3 tools/run_config.py $c --frames 30 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(' ', d['config'], d['kernel_ms'], end='')"
  done
  echo
done
done
