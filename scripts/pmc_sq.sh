#!/bin/bash
# usage: scripts/pmc_sq.sh TAG CONFIG   -- three rocprofv3 --pmc passes (counters only) of a short bench run: wave-cycle shares, instruction mix,
# LDS and instruction-cache counters of every ilqr kernel; summaries in gpurun_out/sq_TAG.txt
TAG=$1; CFG=$2
export PROF_ARGS="--config $CFG"
{
  echo "# $CFG: rocprofv3 --pmc, per-dispatch averages (SQ cycle counters in units of 4 clocks)"
  bash scripts/pmc_brief.sh ${TAG}_a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" &&
  bash scripts/pmc_brief.sh ${TAG}_b "SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" &&
  bash scripts/pmc_brief.sh ${TAG}_c "SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC"
} > gpurun_out/sq_$TAG.txt 2>&1
