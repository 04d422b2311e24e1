#!/bin/bash
# HBM-side byte counters of the four pf::conv_gemm_kernel launches of the stem (4096 events x 3 detectors, bf16).
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_stem
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for P in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -- \
      python3 $GRAFT_REPO_ROOT/scripts/prof_stem.py > $OUT/p$i.log 2>&1
  echo "pass $i rc=$? : $P"
done
python3 $GRAFT_REPO_ROOT/scripts/pmc_summary.py $OUT conv_gemm_kernel | tee $OUT/summary.txt
