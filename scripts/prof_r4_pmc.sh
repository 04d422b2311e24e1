#!/bin/bash
# Round-4 PMC passes + kernel stats (on the GPU box, from the repo root): mixer, the forward kernels (16-row at 4096, wide at
# 65536), the incremental / D-pass inverse, and the bench command's kernel stats + HBM traffic passes.
set -u
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
bash $R/scripts/prof_counters_mixer.sh mixer_r04 > $R/gpurun_out/pmc_mixer_r04.log 2>&1
bash $R/scripts/prof_counters_r2.sh fwd_r04 > $R/gpurun_out/pmc_fwd_r04.log 2>&1
bash $R/scripts/prof_counters_inc.sh inc_r04 > $R/gpurun_out/pmc_inc_r04.log 2>&1
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_r4b
mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $OUT/bench.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sampling -- python3 $R/scripts/prof_sampling.py > $OUT/sampling.log 2>&1
for P in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/traffic_$P -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras > $OUT/traffic_$P.log 2>&1
done
cd $R
for d in bench sampling; do
  f=$(find $OUT/$d -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $OUT/${d}_kernel_stats.csv
done
python3 - <<PY > $OUT/traffic.json
import csv, glob, json, collections
acc = collections.defaultdict(list)
for P in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$OUT/traffic_%s/**/*counter_collection.csv" % P, recursive=True):
        for row in csv.DictReader(open(f)):
            if "flow_kernel" in row.get("Kernel_Name", "") and row["Counter_Name"] == P:
                acc[P].append(float(row["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items() if v}
# FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts 64-byte requests as 32 (MI355X_MICROARCH.md): doubled
out = {"FETCH_SIZE_KB_mean": m.get("FETCH_SIZE"), "WRITE_SIZE_KB_mean": m.get("WRITE_SIZE"), "launches": {k: len(v) for k, v in acc.items()}}
if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
    out["bytes_per_launch"] = int(2 * m["FETCH_SIZE"] * 1024 + m["WRITE_SIZE"] * 1024)
print(json.dumps(out))
PY
rm -rf $OUT/bench $OUT/sampling $OUT/traffic_FETCH_SIZE $OUT/traffic_WRITE_SIZE
