#!/bin/bash
# A/B: default build vs tagged builds; prints avg trace-kernel ms for C2 (fast/strict, LDS vs scalar geometry)
cd "$(dirname "$0")/.."
for lib in "" $@; do
  for variant in 0 1; do
    for strict in 0 1; do
      if [ -n "$lib" ]; then export CLWRAP_LIB=$PWD/example_gui_opencl_raytracer_amd/libopencl_wrap_hip$lib.so; else unset CLWRAP_LIB; fi
      echo -n "lib='$lib' variant=$variant strict=$strict: "
      python3 tools/prof_c2.py 30 $strict $variant | tail -1
    done
  done
done
