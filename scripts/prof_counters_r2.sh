#!/bin/bash
# Round-2 PMC passes for the forward kernels of the bench workload (16-row kernel at B = 4096, wide kernel at B = 65536).
# One rocprofv3 run per counter group (--pmc with --kernel-trace only, as the pool requires).
# usage (on the GPU box): scripts/prof_counters_r2.sh <tag>
set -u
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PASSES=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE SQ_WAVES"
 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -- \
      python3 $GRAFT_REPO_ROOT/scripts/r2_pmc_driver.py > $OUT/p$i.log 2>&1
  echo "pass $i rc=$? : $P"
done
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/scripts/r2_pmc_driver.py > $OUT/trace.log 2>&1
python3 $GRAFT_REPO_ROOT/scripts/pmc_summary.py $OUT | tee $OUT/summary.txt
python3 - <<PY | tee -a $OUT/summary.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "flow_" in row["Kernel_Name"]:
            acc[row["Kernel_Name"].split("(")[0][-70:]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
for k, v in acc.items():
    v = sorted(v)
    print(f"kernel-trace: {k}: n={len(v)} avg {sum(v) / len(v):.1f} us  median {v[len(v) // 2]:.1f} us  min {v[0]:.1f} us")
PY
