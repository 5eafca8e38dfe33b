"""Per-kernel totals of the counters in rocprofv3 rocpd databases (one --pmc pass each): for every kernel the number of
dispatches, the summed kernel time and, per counter, the sum over dispatches and the average per dispatch.
usage: python scripts/pmc_summary.py <out.txt> <db> [<db> ...]"""
import glob
import os
import sqlite3
import sys
from collections import defaultdict


def dbs(path):
    if os.path.isdir(path):
        return sorted(glob.glob(os.path.join(path, '**', '*.db'), recursive=True))
    return [path]


def main():
    out = open(sys.argv[1], 'w')
    for arg in sys.argv[2:]:
        for db in dbs(arg):
            con = sqlite3.connect(db); cur = con.cursor()
            try:
                rows = list(cur.execute("select kernel_name, counter_name, dispatch_id, sum(value), max(end-start) from counters_collection "
                                        "group by kernel_name, counter_name, dispatch_id"))
            except Exception as e:
                out.write(f"# {db}: {e}\n"); continue
            agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0, 0.0]))
            for k, c, d, v, dur in rows:
                a = agg[k][c]; a[0] += 1; a[1] += v; a[2] += dur / 1e6
            out.write(f"# {os.path.basename(os.path.dirname(db))}/{os.path.basename(db)}\n")
            for k in sorted(agg, key=lambda k: -max(a[2] for a in agg[k].values())):
                for c in sorted(agg[k]):
                    n, tot, ms = agg[k][c]
                    out.write("%-70s %-28s n=%5d  sum %.6e  avg %.6e  kernel_ms_total %.3f\n" % (k[:70], c, n, tot, tot / n, ms))
    out.close()


if __name__ == '__main__':
    main()
