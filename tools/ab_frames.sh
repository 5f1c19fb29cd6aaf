#!/bin/bash
# per-frame kernel times of tagged library builds; usage: tools/ab_frames.sh cfg lib1 lib2 ...
cd "$(dirname "$0")/.."
cfg=$1; shift
for round in 1 2 3; do
for lib in "$@"; do
  if [ -n "$lib" ]; then export CLWRAP_LIB=$PWD/example_gui_opencl_raytracer_amd/libopencl_wrap_hip$lib.so; else unset CLWRAP_LIB; fi
  python3 tools/frame_times.py $cfg --frames 24 2>/dev/null | python3 -c "
import sys,json,statistics; d=json.loads(sys.stdin.read()); m=d['ms']; print('lib=%-6s' % '$lib', 'first', m[:4], 'median of rest', round(statistics.median(m[4:]),3), 'max', max(m[4:]))"
done
done
