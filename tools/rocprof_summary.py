#!/usr/bin/env python
"""Summaries of rocprofv3 CSV output for profiles/ (tracked):

  rocprof_summary.py stats  DIR OUT.csv     per-kernel calls / total / average duration from *_kernel_trace.csv
                                            (the same numbers `rocprofv3 --kernel-trace --stats` prints)
  rocprof_summary.py pmc    DIR OUT.json    per-kernel mean of every counter in *_counter_collection.csv, DIR may be
                                            given several times (one per --pmc pass)

Kernel names are shortened to the function name with its template arguments.
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    m = re.match(r"(?:void )?([A-Za-z_0-9:]+(?:<[^(]*>)?)\(", name)
    s = m.group(1) if m else name.split("(")[0]
    return s[-90:]


def stats(root, out):
    agg = collections.OrderedDict()
    for f in sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            a = agg.setdefault(short(r["Kernel_Name"]), [0, 0.0, 1e30, 0.0])
            a[0] += 1
            a[1] += d
            a[2] = min(a[2], d)
            a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values()) or 1.0
    with open(out, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats summary (durations in microseconds), source dir: %s\n" % os.path.basename(root.rstrip("/")))
        f.write("kernel,calls,total_us,avg_us,min_us,max_us,percent\n")
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write('"%s",%d,%.3f,%.3f,%.3f,%.3f,%.3f\n' % (k, a[0], a[1], a[1] / a[0], a[2], a[3], 100 * a[1] / tot))
    print("wrote", out, len(agg), "kernels")


def pmc(roots, out):
    agg = collections.OrderedDict()
    for root in roots:
        for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
            for r in csv.DictReader(open(f)):
                k = agg.setdefault(short(r["Kernel_Name"]), collections.OrderedDict())
                c = k.setdefault(r["Counter_Name"], [0, 0.0])
                c[0] += 1
                c[1] += float(r["Counter_Value"])
    # kernel durations of the SAME passes (rocprofv3 --kernel-trace beside --pmc): the clock a kernel ran at = GRBM_GUI_ACTIVE / 8 XCDs / duration
    dur = collections.OrderedDict()
    for root in roots:
        for f in sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)):
            for r in csv.DictReader(open(f)):
                d = dur.setdefault(short(r["Kernel_Name"]), [0, 0.0])
                d[0] += 1
                d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    res = collections.OrderedDict()
    steps = os.environ.get("VXM_PROFILED_STEPS")
    if steps:                  # bench steps + warm-up steps of the profiled command: bench.py checks dispatch counts against it
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "voxelmorph_amd"))
        import buildinfo                # kernel sources these counters belong to: bench.py refuses them for any other tree
        res["_meta"] = {"steps_profiled": int(steps), "csrc_sha": buildinfo.csrc_sha()}
    for k, ctrs in agg.items():
        res[k] = {c: {"dispatches": v[0], "mean": v[1] / v[0]} for c, v in ctrs.items()}
    for k, v in res.items():
        # MFMA-pipe utilisation: SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs (= 32 cycles x SQ_INSTS_MFMA for
        # v_mfma_f32_16x16x4_f32), GRBM_GUI_ACTIVE over the 8 XCDs (checked against kernel duration x 2.4 GHz)
        if k != "_meta" and "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v and v["GRBM_GUI_ACTIVE"]["mean"] > 0:
            v["mfma_util"] = v["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / 1024.0 / (v["GRBM_GUI_ACTIVE"]["mean"] / 8.0)
        if k != "_meta" and "GRBM_GUI_ACTIVE" in v and k in dur and dur[k][1] > 0:
            v["duration_us_under_pmc"] = dur[k][1] / dur[k][0]
            v["shader_clock_ghz"] = v["GRBM_GUI_ACTIVE"]["mean"] / 8.0 / (v["duration_us_under_pmc"] * 1e3)
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print("wrote", out, len(res), "kernels")


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2:-1], sys.argv[-1])
