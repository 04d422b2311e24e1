#!/usr/bin/env python3
"""Average the per-dispatch PMC values of the flow kernels out of rocprofv3 CSV passes."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
match = sys.argv[2] if len(sys.argv) > 2 else "flow_"
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True)):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")
            if match not in name:
                continue
            short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void pf::", "")[-60:]
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(f"== {k}")
    for c, v in sorted(d.items()):
        print(f"  {c:34s} mean {sum(v) / len(v):16.1f}   n={len(v)}")
