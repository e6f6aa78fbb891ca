"""Timeline of the last N kernels of a rocprofv3 rocpd database: start offset, duration and the
idle gap before each launch (development: host / launch overhead inside one MPC step)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
rows = rows[-n:]
t0 = rows[0][1]
prev_end = None
busy = gap = 0
for name, s, e in rows:
    g = (s - prev_end) if prev_end is not None else 0
    short = name.split("(")[0].replace("void agx::", "")[:40]
    print(f"{(s - t0) / 1e3:10.1f}us  dur {(e - s) / 1e3:8.1f}us  gap {g / 1e3:7.1f}us  {short}")
    busy += e - s
    gap += max(g, 0)
    prev_end = max(e, prev_end or e)
print(f"busy {busy / 1e3:.1f}us  idle {gap / 1e3:.1f}us")
