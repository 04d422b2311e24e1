#!/bin/bash
# SQ counter passes for the stem's pf::conv_gemm_kernel launches (4096 events x 3 detectors, bf16); one rocprofv3 run per pass.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_stem_sq${1:+_$1}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PASSES=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
 "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE SQ_WAVES"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -- \
      python3 $GRAFT_REPO_ROOT/scripts/prof_stem.py > $OUT/p$i.log 2>&1
  echo "pass $i rc=$? : $P"
done
python3 $GRAFT_REPO_ROOT/scripts/pmc_summary.py $OUT conv_ | tee $OUT/summary.txt
