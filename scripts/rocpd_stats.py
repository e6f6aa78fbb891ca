"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) as a --stats-like table."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {name_col} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"{'kernel':90s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
for n, c, s, a, mn, mx in rows:
    print(f"{n[:90]:90s} {c:6d} {s/1e3:12.1f} {a/1e3:10.2f} {mn/1e3:10.2f} {mx/1e3:10.2f} {100*s/tot:6.2f}")
