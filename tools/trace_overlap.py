#!/usr/bin/env python
"""Step anatomy from a rocprofv3 --kernel-trace CSV: for the last N steps (a step ends with the Adam kernel) the span on the device, the sum
of kernel durations, the time two or more kernels ran side by side, the idle gaps, and the kernels per queue.
usage: trace_overlap.py DIR [nsteps | first last]   (steps are numbered by their Adam launch, 0 = the first step of the process)"""
import csv, glob, os, sys

dump = None
if "--dump" in sys.argv:
    i = sys.argv.index("--dump")
    dump = sys.argv[i + 1]
    del sys.argv[i:i + 2]
root = sys.argv[1]
nsteps = int(sys.argv[2]) if len(sys.argv) == 3 else 3
span = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else None
rows = []
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
ends = [i for i, r in enumerate(rows) if "k_adam" in r[2] and "tick" not in r[2]]
print("dispatches", len(rows), "adam launches", len(ends))
for k in (range(max(1, span[0]), min(span[1] + 1, len(ends))) if span else range(max(1, len(ends) - nsteps), len(ends))):
    seg = rows[ends[k - 1] + 1: ends[k] + 1]
    t0, t1 = seg[0][0], max(r[1] for r in seg)
    ev = sorted([(r[0], 1) for r in seg] + [(r[1], -1) for r in seg])
    busy1 = busy2 = 0
    depth, last = 0, t0
    for t, d in ev:
        if depth >= 1: busy1 += t - last
        if depth >= 2: busy2 += t - last
        depth += d; last = t
    queues = {}
    for r in seg:
        queues[r[3]] = queues.get(r[3], 0) + 1
    gaps = sorted(((seg[i + 1][0] - max(r[1] for r in seg[:i + 1])) / 1e3 for i in range(len(seg) - 1)), reverse=True)
    if dump:
        with open(dump, "a") as fh:
            fh.write("== step %d\n" % k)
            for r in seg:
                name = r[2].replace("(anonymous namespace)::", "").split("(")[0][-60:]
                fh.write("%10.1f us  %8.1f us  queue %-3s %s\n" % ((r[0] - t0) / 1e3, (r[1] - r[0]) / 1e3, r[3], name))
    print("step %d: %d kernels, span %.3f ms, sum of durations %.3f ms, >=1 kernel %.3f ms, >=2 kernels %.3f ms, idle %.3f ms, queues %s, largest gaps (us) %s"
          % (k, len(seg), (t1 - t0) / 1e6, sum(r[1] - r[0] for r in seg) / 1e6, busy1 / 1e6, busy2 / 1e6, (t1 - t0 - busy1) / 1e6, queues,
             ["%.1f" % g for g in gaps[:5]]))
