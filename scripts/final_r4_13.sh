#!/bin/bash
# after R4.13 (incremental inverse skips all-zero fragment groups): whole GPU suite, bench line, sampling kernel trace
set -e
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4_13; mkdir -p $OUT; REPO=$PWD
python -m pytest tests -m gpu -x -q > $OUT/gputests.log 2>&1
python bench.py > $OUT/bench.log 2>&1
cd /tmp
PYTHONPATH=$REPO rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/samp -- python3 $REPO/scripts/prof_sampling.py > $OUT/samp.log 2>&1
cd $REPO
cp $(find $OUT/samp -name '*kernel_stats.csv' | head -1) $OUT/sampling_kernel_stats.csv
rm -rf $OUT/samp
tail -2 $OUT/gputests.log; head -4 $OUT/sampling_kernel_stats.csv | cut -c1-160
