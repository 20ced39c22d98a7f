#!/usr/bin/env python3
"""Per-kernel table of one bench.py step from three rocprofv3 runs of the SAME command (rocpd sqlite databases):

    rocprofv3 --kernel-trace --stats -d T -- python3 bench.py ...      durations
    rocprofv3 --pmc FETCH_SIZE      -d F -- python3 bench.py ...      HBM-side read traffic   (own pass: TCC slots)
    rocprofv3 --pmc WRITE_SIZE      -d W -- python3 bench.py ...      HBM-side write traffic  (own pass)

    python tools/pmc_table.py T/*/*.db F/*/*.db W/*/*.db STEPS > profiles/rNN_pmc.csv

STEPS = steps the command ran (warm-up + timed), to turn call counts into launches per step.  Kernels are keyed by
(name, grid size) so that the shapes of one template show up separately.  traffic_bytes applies the corrections of
MI355X_MICROARCH.md (HBM section): FETCH_SIZE is reported in KiB and, on gfx950, at half the bytes of wide streaming
reads -> x 2 x 1024; WRITE_SIZE KiB -> x 1024.  bench.py reads this file for `roofline.traffic`.
"""
import csv
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*\)$", "", name)          # drop the argument list
    return name


def durations(path):
    c = sqlite3.connect(path)
    out = {}
    for name, grid, n, avg in c.execute("select name, grid_x * grid_y * grid_z, count(*), avg(end - start) from kernels group by 1, 2"):
        out[(short(name), int(grid))] = (n, avg / 1e3)
    return out


def counter(path, which):
    c = sqlite3.connect(path)
    out = {}
    for name, grid, n, avg in c.execute("select kernel_name, grid_size, count(*), avg(value) from counters_collection "
                                        "where counter_name = ? group by kernel_name, grid_size", (which,)):
        out[(short(name), int(grid))] = (n, avg)
    return out


def main():
    trace, fetch, write, steps = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
    d, f, w = durations(trace), counter(fetch, "FETCH_SIZE"), counter(write, "WRITE_SIZE")
    rows = []
    for key, (n, us) in d.items():
        if n < steps * 0.5:          # set-up kernels (initialisation, casts before the loop)
            continue
        fk, wk = f.get(key, (0, float("nan")))[1], w.get(key, (0, float("nan")))[1]
        rows.append((n / steps * us, key[0], key[1], n / steps, us, fk, wk, (2.0 * fk + wk) * 1024.0))
    rows.sort(reverse=True)
    out = csv.writer(sys.stdout)
    out.writerow(["kernel", "grid_size", "launches_per_step", "avg_us", "ms_per_step", "FETCH_SIZE_KiB_avg", "WRITE_SIZE_KiB_avg",
                  "traffic_bytes_per_launch"])
    for ms, name, grid, per, us, fk, wk, tb in rows:
        out.writerow([name, grid, "%.2f" % per, "%.2f" % us, "%.4f" % (ms / 1e3), "%.1f" % fk, "%.1f" % wk, "%.0f" % tb])


if __name__ == "__main__":
    main()
