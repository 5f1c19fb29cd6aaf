#!/bin/bash
# rocprofv3 evidence for ONE BASELINE configuration through tools/run_config.py (kernel stats + the PMC passes, each in its
# own run, never combined with tracing) into gpurun_out/prof_$1_$2; condense with tools/summarize_prof.py.
#   usage: tools/profile_config.sh <tag> <c2|c3|c4|ref800> [frames]
tag=$1; cfg=$2; frames=${3:-12}
out=gpurun_out/prof_${tag}_${cfg}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python3 tools/run_config.py $cfg --frames $frames"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $CMD > $out/trace.log 2>&1 &&
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $out/pmc_sq1 -- $CMD > $out/pmc_sq1.log 2>&1 &&
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA --output-format csv -d $out/pmc_sq2 -- $CMD > $out/pmc_sq2.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 --output-format csv -d $out/pmc_mix -- $CMD > $out/pmc_mix.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $CMD > $out/pmc_fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- $CMD > $out/pmc_write.log 2>&1 &&
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_l2 -- $CMD > $out/pmc_l2.log 2>&1
echo "profile_config $cfg rc=$?"
grep -h '"config"' $out/trace.log | cut -c1-400
