#!/bin/bash
mkdir -p gpurun_out
for lib in "" _prev "" _prev; do
  if [ -n "$lib" ]; then export CLWRAP_LIB=$PWD/example_gui_opencl_raytracer_amd/libopencl_wrap_hip$lib.so; else unset CLWRAP_LIB; fi
  python3 tools/run_config.py c3 --frames 8 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib=$lib c3', d['kernel_ms'], d['Mrays_s'])"
done
for lib in "" _prev; do
  if [ -n "$lib" ]; then export CLWRAP_LIB=$PWD/example_gui_opencl_raytracer_amd/libopencl_wrap_hip$lib.so; else unset CLWRAP_LIB; fi
  python3 tools/run_config.py c3 --frames 6 --strict 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib=$lib c3 strict', d['kernel_ms'], d['Mrays_s'])"
  python3 tools/run_config.py c2 --frames 100 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib=$lib c2', d['kernel_ms'], d['Mrays_s'])"
  timeout -k 10 600 python tools/tpt_check.py time ref800,hd15 48 2>&1 | cut -c1-100
done
