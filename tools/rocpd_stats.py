#!/usr/bin/env python
"""Per-kernel statistics from a rocprofv3 rocpd sqlite database (the default output format of
rocprofv3 in ROCm 7.2) - same columns as `--stats` kernel_stats.csv.  Usage: rocpd_stats.py run_results.db"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in cur.execute("pragma table_info(%s)" % ks)]
    name_col = "kernel_name" if "kernel_name" in cols else ("display_name" if "display_name" in cols else cols[-1])
    rows = cur.execute("select s.%s, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
                       "from %s d join %s s on d.kernel_id = s.id group by s.%s order by 3 desc" % (name_col, kd, ks, name_col)).fetchall()
    tot = float(sum(r[2] for r in rows)) or 1.0
    print("%-90s %8s %14s %12s %12s %12s %7s" % ("Name", "Calls", "TotalNs", "AvgNs", "MinNs", "MaxNs", "Pct"))
    for n, c, t, a, mn, mx in rows:
        print("%-90s %8d %14d %12.0f %12d %12d %6.2f%%" % (str(n)[:90], c, t, a, mn, mx, 100.0 * t / tot))


if __name__ == "__main__":
    main(sys.argv[1])
