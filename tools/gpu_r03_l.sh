#!/bin/bash
mkdir -p gpurun_out
export CLWRAP_LIB=$PWD/example_gui_opencl_raytracer_amd/libopencl_wrap_hip_tl.so
for cfg in ref800 c3; do for k in 24 48; do timeout -k 10 120 python tools/timeline.py $cfg --tpt $k 2>&1 | tail -1 | cut -c1-800; done; done | tee gpurun_out/tpt_timeline.log
