"""Per-kernel PMC counter sums from a rocprofv3 --pmc run (rocpd sqlite)."""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
rows = cur.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall() if "kernel_name" in cols else []
raw = cur.execute("select kernel_name, counter_name, value from counters_collection").fetchall() if "kernel_name" in cols else []
if not rows:
    print(cols); sys.exit()
agg = collections.defaultdict(dict)
for k, c, v, n in rows:
    agg[k][c] = (v, n)
for k, d in agg.items():
    if len(sys.argv) > 2 and sys.argv[2] not in k: continue
    n = max(x[1] for x in d.values())
    print(k[:70], "dispatches~", n)
    for c, (v, cnt) in sorted(d.items()):
        print(f"   {c:32s} total {v:16.0f}  per-dispatch {v/max(cnt,1):14.1f}")
