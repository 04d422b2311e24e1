"""Per-kernel time of the LAST `ms` milliseconds of a rocprofv3 kernel trace (steady state, behind MIOpen's search)."""
import csv, sys, collections
path, ms = sys.argv[1], float(sys.argv[2])
rows = list(csv.DictReader(open(path)))
end = max(int(r["End_Timestamp"]) for r in rows)
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    if int(r["Start_Timestamp"]) >= end - ms * 1e6:
        a = agg[r["Kernel_Name"]]; a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in agg.values())
print(f"window {ms} ms: kernel time {tot / 1e6:.2f} ms in {sum(v[0] for v in agg.values())} launches")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"{v[0]:5d} calls {v[1] / 1e3:9.1f} us  {k[:130]}")
