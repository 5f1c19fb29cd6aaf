#!/bin/bash
# Collects the rocprofv3 evidence for one round into gpurun_out/prof_$1 (copy the summaries to profiles/).
#   pass 1: --kernel-trace --stats of the bench command        -> per-kernel average duration
#   pass 2..: --pmc in separate runs (never combined with tracing): SQ mix / stalls, VALU instruction classes, FETCH_SIZE, WRITE_SIZE
tag=${1:-r01}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BENCH="python3 bench.py --gpus 1 --steps 50 --warmup 5 --no-cpu-baseline --no-strict-leg --legs none"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $BENCH > $out/trace.log 2>&1 &&
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $out/pmc_sq1 -- $BENCH > $out/pmc_sq1.log 2>&1 &&
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA --output-format csv -d $out/pmc_sq2 -- $BENCH > $out/pmc_sq2.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 --output-format csv -d $out/pmc_mix -- $BENCH > $out/pmc_mix.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $BENCH > $out/pmc_fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- $BENCH > $out/pmc_write.log 2>&1 &&
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_l2 -- $BENCH > $out/pmc_l2.log 2>&1
echo "profile_round rc=$?"
grep -h '"metric"' $out/*.log | cut -c1-200
