#!/bin/bash
timeout -k 10 300 python tools/tpt_check.py quick > gpurun_out/tpt_quick.log 2>&1; echo "quick rc=$?"; tail -1 gpurun_out/tpt_quick.log
for lib in _l4 "" _l6 _l4 "" _l6; do
  if [ -n "$lib" ]; then export CLWRAP_LIB=$PWD/example_gui_opencl_raytracer_amd/libopencl_wrap_hip$lib.so; else unset CLWRAP_LIB; fi
  echo "lib=$lib"; timeout -k 10 600 python tools/tpt_check.py time ref800,hd15,c3s 40 2>&1 | cut -c1-100
done
