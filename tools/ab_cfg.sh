#!/bin/bash
# A/B of tagged library builds on one run_config configuration: tools/ab_cfg.sh <config> <extra args> -- "" _tag ...
cd "$(dirname "$0")/.."
cfg=$1; shift; extra=""
while [ "$1" != "--" ]; do extra="$extra $1"; shift; done; shift
for round in 1 2; do
for lib in "$@"; do
  if [ -n "$lib" ]; then export CLWRAP_LIB=$PWD/example_gui_opencl_raytracer_amd/libopencl_wrap_hip$lib.so; else unset CLWRAP_LIB; fi
  echo -n "lib='$lib': "
  python3 tools/run_config.py $cfg $extra 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernel_ms'], d['Mrays_s'])"
done
done
