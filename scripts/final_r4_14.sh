#!/bin/bash
# closing kernel traces of round 4: the sampling kernels after the zero-group skip, and the default bench command
set -e
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4_14; mkdir -p $OUT; REPO=$PWD
cd /tmp
PYTHONPATH=$REPO rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/samp -- python3 $REPO/scripts/prof_sampling.py > $OUT/samp.log 2>&1
cp $(find $OUT/samp -name '*kernel_stats.csv' | head -1) $OUT/sampling_kernel_stats.csv
rm -rf $OUT/samp
PYTHONPATH=$REPO rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -- python3 $REPO/bench.py > $OUT/bench.log 2>&1
cp $(find $OUT/bench -name '*kernel_stats.csv' | head -1) $OUT/bench_kernel_stats.csv
rm -rf $OUT/bench
cd $REPO
head -4 $OUT/sampling_kernel_stats.csv | cut -c1-160; head -4 $OUT/bench_kernel_stats.csv | cut -c1-160
