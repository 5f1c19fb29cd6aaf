#!/bin/bash
cd "$(dirname "$0")/.."
for lib in "$@"; do
  if [ -n "$lib" ]; then export CLWRAP_LIB=$PWD/example_gui_opencl_raytracer_amd/libopencl_wrap_hip$lib.so; else unset CLWRAP_LIB; fi
  echo "lib='$lib':"; python3 tools/lone_wave.py 2>/dev/null | grep "variant 0" | cut -c1-60,100-200
done
