"""Summarise a rocprofv3 rocpd sqlite database (--kernel-trace) into a per-kernel stats table (like --stats csv)."""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
rows = list(cur.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e6, min(end-start)/1e6, max(end-start)/1e6 "
                        "from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print("%-72s %6s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_ms", "avg_ms", "min_ms", "max_ms", "pct"))
for r in rows:
    print("%-72s %6d %12.2f %10.3f %10.3f %10.3f %6.2f" % (r[0][:72], r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot))
print("total kernel time %.1f ms" % tot)
