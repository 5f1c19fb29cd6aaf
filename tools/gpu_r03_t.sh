#!/bin/bash
for lib in "" _rc8 "" _rc8; do
  if [ -n "$lib" ]; then export CLWRAP_LIB=$PWD/example_gui_opencl_raytracer_amd/libopencl_wrap_hip$lib.so; else unset CLWRAP_LIB; fi
  echo "lib=$lib"; timeout -k 10 600 python tools/tpt_check.py time ref800,hd15,c3s 48 2>&1 | cut -c1-100
done
