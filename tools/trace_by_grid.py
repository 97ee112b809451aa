#!/usr/bin/env python
"""rocprofv3 --kernel-trace CSV -> average duration per (kernel, grid, workgroup): tells the launches of one kernel apart.
    python tools/trace_by_grid.py DIR [name-filter]"""
import collections, csv, glob, os, re, sys
agg = collections.OrderedDict()
for f in sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
        if len(sys.argv) > 2 and sys.argv[2] not in name:
            continue
        key = (name, r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"), r.get("Workgroup_Size_X", "?"))
        a = agg.setdefault(key, [0, 0.0, 1e30])
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a[0] += 1; a[1] += d; a[2] = min(a[2], d)
for (name, gx, gy, wx), (n, tot, mn) in agg.items():
    print("%-44s grid %7s x %3s  wg %4s  calls %4d  avg %8.2f us  min %8.2f" % (name[-44:], gx, gy, wx, n, tot / n, mn))
