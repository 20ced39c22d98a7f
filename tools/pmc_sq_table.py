#!/usr/bin/env python3
"""rocprofv3 --pmc <SQ counters> GRBM_GUI_ACTIVE --kernel-trace --output-format csv  ->  one row per (kernel, grid size):
launches, average duration, effective clock (GRBM_GUI_ACTIVE / 8 XCDs / duration), the counters' means and the fractions
that matter for an MFMA kernel (MI355X_MICROARCH.md, PMC slots: WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES in
quad-cycles; SQ_VALU_MFMA_BUSY_CYCLES in cycles summed over the SIMDs -> / (1024 SIMDs x kernel cycles) = MFMA pipe utilisation).
    python tools/pmc_sq_table.py <rocprofv3 output dir> > table.csv"""
import collections
import csv
import glob
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*\)$", "", name)


def main():
    root = sys.argv[1]
    cnt = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            cnt[(short(r["Kernel_Name"]), int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    for fn in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) if "Grid_Size_X" in r else int(r["Grid_Size"])
            dur[(short(r["Kernel_Name"]), g)].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    names = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_VALU_MFMA_BUSY_CYCLES",
             "SQ_WAIT_INST_LDS", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE"]
    out = csv.writer(sys.stdout)
    out.writerow(["kernel", "grid_size", "launches", "avg_us", "total_ms", "clock_GHz", "mfma_util", "wait_any_frac", "wait_inst_frac",
                  "active_inst_frac", "wait_inst_lds_frac"] + names)
    rows = []
    for key, c in cnt.items():
        d = dur.get(key)
        if not d:
            continue
        m = {k: (sum(v) / len(v) if v else float("nan")) for k, v in c.items()}
        us = sum(d) / len(d) / 1e3
        gui = m.get("GRBM_GUI_ACTIVE", float("nan")) / 8.0
        wc = m.get("SQ_WAVE_CYCLES", float("nan"))
        f = lambda k: m.get(k, float("nan")) / wc if wc else float("nan")
        rows.append((us * len(d), [key[0], key[1], len(d), "%.2f" % us, "%.3f" % (us * len(d) / 1e3), "%.3f" % (gui / (us * 1e3)),
                                   "%.3f" % (m.get("SQ_VALU_MFMA_BUSY_CYCLES", float("nan")) / (1024.0 * gui)), "%.3f" % f("SQ_WAIT_ANY"),
                                   "%.3f" % f("SQ_WAIT_INST_ANY"), "%.3f" % f("SQ_ACTIVE_INST_ANY"), "%.3f" % f("SQ_WAIT_INST_LDS")]
                     + ["%.4g" % m.get(k, float("nan")) for k in names]))
    for _, r in sorted(rows, key=lambda t: -t[0]):
        out.writerow(r)


if __name__ == "__main__":
    main()
