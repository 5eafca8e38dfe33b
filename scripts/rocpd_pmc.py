"""Per-kernel PMC sums from a rocprofv3 rocpd sqlite database (--pmc)."""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
q = "select kernel_name, counter_name, dispatch_id, sum(value), max(end-start) from counters_collection group by kernel_name, counter_name, dispatch_id"
try:
    rows = list(cur.execute(q))
except Exception as e:
    print(cols); raise
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(list))
for k, c, d, v, dur in rows:
    agg[k][c].append((d, v, dur))
for k in agg:
    if len(sys.argv) > 2 and sys.argv[2] not in k: continue
    print(k[:100])
    for c in sorted(agg[k]):
        vals = agg[k][c]
        print("   %-32s n=%3d " % (c, len(vals)), "  ".join("%.4g" % v[1] for v in vals[:6]), " | dur_ms", "  ".join("%.2f" % (v[2] / 1e6) for v in vals[:6]))
