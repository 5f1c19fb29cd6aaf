#!/bin/bash
for lib in "" _prev "" _prev; do
  if [ -n "$lib" ]; then export CLWRAP_LIB=$PWD/example_gui_opencl_raytracer_amd/libopencl_wrap_hip$lib.so; else unset CLWRAP_LIB; fi
  python3 tools/run_config.py c3 --frames 8 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib=$lib c3', d['kernel_ms'], d['Mrays_s'])"
done
