#!/bin/bash
# grid density sweep of C4 per library build; usage: tools/c4_sweep.sh lib...
cd "$(dirname "$0")/.."
for lib in "$@"; do
if [ -n "$lib" ]; then export CLWRAP_LIB=$PWD/example_gui_opencl_raytracer_amd/libopencl_wrap_hip$lib.so; else unset CLWRAP_LIB; fi
for d in 0.8 1.0 1.2 1.4 1.6 1.8 2.0 2.4 3.0; do echo -n "lib=$lib density $d: "; CLWRAP_GRID_DENSITY=$d python3 tools/frame_times.py c4 --frames 16 2>/dev/null | python3 -c "import sys,json,statistics; d=json.loads(sys.stdin.read()); print(round(statistics.median(d['ms'][4:]),3))"; done
done
