#!/bin/bash
# kernel time + VALU/SALU instruction counts of tagged builds: tools/pmc_valu.sh <cfg> lib...
cfg=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  if [ -n "$lib" ]; then export CLWRAP_LIB=$PWD/example_gui_opencl_raytracer_amd/libopencl_wrap_hip$lib.so; else unset CLWRAP_LIB; fi
  out=gpurun_out/pmcv_${cfg}${lib}; rm -rf $out; mkdir -p $out
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $out/pmc_sq1 -- python3 tools/run_config.py $cfg --frames 6 > $out/pmc_sq1.log 2>&1
  echo "=== lib='$lib' $(grep -h '"config"' $out/pmc_sq1.log | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['kernel_ms'])")"
  python3 tools/summarize_prof.py $out $out pmcv | grep -E "^SQ_"
done
