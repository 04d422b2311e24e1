#!/bin/bash
# HBM-side byte counters of pf::remix_kernel (separate rocprofv3 --pmc passes, run on the GPU box via gpurun).
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_remix
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for P in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -- \
      python3 $GRAFT_REPO_ROOT/scripts/bench_remix.py > $OUT/p$i.log 2>&1
  echo "pass $i rc=$? : $P"
done
python3 $GRAFT_REPO_ROOT/scripts/pmc_summary.py $OUT remix_kernel | tee $OUT/summary.txt
