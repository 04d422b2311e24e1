#!/usr/bin/env python3
"""usage: kernel_timeline.py <rocprofv3 *_kernel_trace.csv> [first-kernel substring] -- the LAST iteration (from the last
dispatch whose name contains the substring, default remix_kernel) as a timeline: start offset, duration, queue, gap to the
previous kernel's end on any queue, name."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
key = sys.argv[2] if len(sys.argv) > 2 else "remix_kernel"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
lo = starts[-2] if len(starts) > 1 else starts[-1]
hi = starts[-1] if len(starts) > 1 else len(rows)
t0 = int(rows[lo]["Start_Timestamp"])
end_prev = t0
busy = 0
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = s - end_prev
    print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  q{r['Queue_Id']:>2s}  gap {gap / 1e3:7.1f}  grid {int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])):6d}x{r['Grid_Size_Y']:>3s}x{r['Grid_Size_Z']:>3s}  {r['Kernel_Name'][:90]}")
    end_prev = max(end_prev, e)
print(f"iteration: {(end_prev - t0) / 1e3:.1f} us")
