#!/bin/bash
# Round-4 kernel-trace profiles: training step (1024 events), config 3 end to end + single-event sampling (both precisions).
# usage (on the GPU box, from the repo root): bash scripts/prof_r4.sh
set -e
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_r4
mkdir -p $OUT
REPO=$PWD
export PYTHONPATH=$REPO
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train -- python3 $REPO/scripts/bench_train.py > $OUT/train.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/e2e -- python3 $REPO/scripts/bench_e2e.py > $OUT/e2e.log 2>&1
cd $REPO
for d in train e2e; do
  f=$(find $OUT/$d -name '*kernel_stats.csv' | head -1)
  cp $f $OUT/${d}_kernel_stats.csv
  python3 scripts/kernel_stats_summary.py $f 1 40 > $OUT/${d}_summary.txt
  grep -c Cijk $f > $OUT/${d}_cijk_count.txt || true
done
t=$(find $OUT/train -name '*kernel_trace.csv' | head -1)
python3 scripts/kernel_timeline.py $t > $OUT/train_timeline.txt 2>&1 || true
rm -rf $OUT/train $OUT/e2e
