#!/usr/bin/env python
"""Summarise rocprofv3 --pmc CSV output (counter_collection.csv files under a directory): one line per
(kernel, dispatch) with every counter collected, kernel names shortened.  Usage: pmc_summary.py DIR [filter]"""
import collections
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rocprof_summary import short as _short  # noqa: E402

root = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.OrderedDict()
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    seen = collections.Counter()
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if flt and flt not in name:
            continue
        short = _short(name)[:60]
        agg.setdefault((short, r["Grid_Size"]), collections.OrderedDict()).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for (short, grid), ctrs in agg.items():
    print("%s grid=%s" % (short, grid))
    for c, vals in ctrs.items():
        print("    %-40s n=%d mean=%.4g" % (c, len(vals), sum(vals) / len(vals)))
        if os.environ.get("PMC_PER_DISPATCH") and c in ("FETCH_SIZE", "WRITE_SIZE", "GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES"):
            print("        per dispatch: " + " ".join("%.4g" % v for v in vals))
# kernel durations from the kernel trace, if present
for f in sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True))[:1]:
    dur = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if flt and flt not in name:
            continue
        short = _short(name)[:60]
        dur.setdefault((short, r.get("Grid_Size", r.get("Grid_Size_X", "?"))), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for (short, grid), v in dur.items():
        print("%s grid=%s  n=%d  avg %.1f us  min %.1f us" % (short, grid, len(v), sum(v) / len(v), min(v)))
