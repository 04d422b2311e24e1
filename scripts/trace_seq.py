"""Launch sequence of the last `ms` milliseconds of a rocprofv3 kernel trace (+ memory copies if traced)."""
import csv, sys
path, ms = sys.argv[1], float(sys.argv[2])
rows = list(csv.DictReader(open(path)))
end = max(int(r["End_Timestamp"]) for r in rows)
sel = sorted((r for r in rows if int(r["Start_Timestamp"]) >= end - ms * 1e6), key=lambda r: int(r["Start_Timestamp"]))
t0 = int(sel[0]["Start_Timestamp"])
for r in sel:
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} us +{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f}  {r['Kernel_Name'][:110]}")
