#!/bin/bash
# PMC passes for the backward kernels (chain, re-evaluation, training forward) of scripts/prof_flow_bwd.py (bf16 mode,
# 2048 rows).  One rocprofv3 run per counter group (--pmc with --kernel-trace only, as the pool requires).
# usage (on the GPU box): scripts/prof_counters_bwd.sh <tag>
set -u
TAG=${1:-bwd}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PASSES=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"
 "TCC_HIT_sum TCC_MISS_sum"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/pmc_$TAG/p$i -- \
      python3 $GRAFT_REPO_ROOT/scripts/prof_flow_bwd.py bf16 > $OUT/p$i.log 2>&1
  echo "pass $i rc=$? : $P"
done
python3 $GRAFT_REPO_ROOT/scripts/pmc_summary.py /tmp/pmc_$TAG flow_ | tee $OUT/summary.txt
