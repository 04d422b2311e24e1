#!/usr/bin/env python3
"""usage: kernel_stats_summary.py <rocprofv3 *_kernel_stats.csv> [iterations] -- per-kernel totals per iteration"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
it = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total kernel time per iteration: {tot / it / 1e6:.3f} ms ({int(sum(int(r['Calls']) for r in rows) / it)} launches)")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    print(f"{float(r['TotalDurationNs']) / it / 1e3:9.1f} us {int(r['Calls']) / it:6.1f} x {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:110]}")
