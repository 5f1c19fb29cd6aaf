#!/bin/bash
mkdir -p gpurun_out
export CLWRAP_LIB=$PWD/example_gui_opencl_raytracer_amd/libopencl_wrap_hip_tl.so
for k in 0 24 48 64; do timeout -k 10 120 python tools/timeline.py ref800 --tpt $k 2>&1 | tail -1 | cut -c1-700; done | tee gpurun_out/tpt_timeline.log
unset CLWRAP_LIB
for cfg in ref800 c3; do
  bash tools/ab_cfg.sh $cfg --variant 16 -- "" _old 2>&1 | tee -a gpurun_out/tpt_ab.log
done
