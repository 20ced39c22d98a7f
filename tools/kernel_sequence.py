#!/usr/bin/env python3
"""Launch order of the kernels of the LAST step in a rocprofv3 kernel-trace database (rocpd sqlite), one short line per launch:
finds who issues the small runtime kernels (copyBuffer / fillBuffer) by their neighbours.
    python tools/kernel_sequence.py gpurun_out/<run>/trace/*/*.db [n_last]"""
import re
import sqlite3
import sys


def main(path, n_last=500):
    c = sqlite3.connect(path)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    rows = rows[-n_last:]
    t0 = rows[0][1]
    for name, s, e in rows:
        short = re.sub(r"\(anonymous namespace\)::", "", name)
        short = re.sub(r"\(.*", "", short)[:90]
        print("%10.1f us  %7.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, short))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 500)
