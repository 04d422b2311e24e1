"""Summarise a rocprofv3 rocpd database: top kernels by total time (name, calls, total ms, %)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows = list(db.execute("select name, total_calls, total_duration, percentage from top_kernels"))
tot = sum(r[2] for r in rows)
print(f"total kernel time {tot/1e3:.2f} ms over {sum(r[1] for r in rows)} launches")
for name, calls, dur, pct in rows[:n]:
    print(f"{dur/1e3:9.3f} ms {pct:5.1f}% {calls:6d}x  {name[:110]}")
