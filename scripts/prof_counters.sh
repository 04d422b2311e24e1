#!/bin/bash
# PMC passes for the flow forward kernel (run on the GPU box via gpurun).
# Each pass is its own rocprofv3 run (--pmc with --kernel-trace only), as the pool requires.
# usage: scripts/prof_counters.sh <tag> [extra bench args]
set -u
TAG=${1:-run}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PASSES=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
 "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE SQ_WAVES"
 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -- \
      python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-graph "$@" > $OUT/p$i.log 2>&1
  echo "pass $i rc=$? : $P"
done
python3 $GRAFT_REPO_ROOT/scripts/pmc_summary.py $OUT | tee $OUT/summary.txt
