#!/bin/bash
# PMC A/B of tagged library builds on one configuration (SQ instruction / wait counters + vector-cache counters, each pass its
# own run).  usage: tools/pmc_ab.sh <cfg> lib1 lib2 ...   -> gpurun_out/pmcab_<cfg><lib>/ and a condensed summary on stdout
cfg=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  if [ -n "$lib" ]; then export CLWRAP_LIB=$PWD/example_gui_opencl_raytracer_amd/libopencl_wrap_hip$lib.so; else unset CLWRAP_LIB; fi
  out=gpurun_out/pmcab_${cfg}${lib}
  rm -rf $out; mkdir -p $out
  CMD="python3 tools/run_config.py $cfg --frames 8"
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $out/pmc_sq1 -- $CMD > $out/pmc_sq1.log 2>&1 &&
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA --output-format csv -d $out/pmc_sq2 -- $CMD > $out/pmc_sq2.log 2>&1 &&
  rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum --output-format csv -d $out/pmc_tcp -- $CMD > $out/pmc_tcp.log 2>&1 &&
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum --output-format csv -d $out/pmc_l2 -- $CMD > $out/pmc_l2.log 2>&1
  echo "=== lib='$lib' rc=$?"
  python3 tools/summarize_prof.py $out $out pmcab | grep -v "^$" | grep -A40 "### kernel"
done
