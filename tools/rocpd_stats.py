#!/usr/bin/env python3
"""rocprofv3 (ROCm 7.2) writes a rocpd sqlite database; this dumps the per-kernel summary that `--stats` describes
(name, calls, total / average / min / max duration in ns, share of total kernel time) as CSV.

    python tools/rocpd_stats.py gpurun_out/<run>/prof/*/*_results.db > profiles/<name>_kernel_stats.csv
"""
import csv
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
                     "from kernels group by name order by 3 desc").fetchall()
    total = float(sum(r[2] for r in rows)) or 1.0
    w = csv.writer(sys.stdout)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for name, n, tot, avg, mn, mx in rows:
        w.writerow([name, n, tot, "%.1f" % avg, "%.2f" % (100.0 * tot / total), mn, mx])


if __name__ == "__main__":
    main(sys.argv[1])
