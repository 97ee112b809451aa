#!/bin/bash
# GPU box helper: kernel trace of 3 bench steps; lists every dispatch of kernels matching FILTER with grid size and duration.
# usage: tools/trace_step.sh OUT FILTER [bench args]
set -u
OUT=$1; FLT=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf "$OUT"; mkdir -p "$OUT"
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/bench.log 2>&1
python - "$OUT" "$FLT" <<'PY'
import csv, glob, sys, os
out, flt = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(os.path.join(out, "t", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size", r.get("Grid_Size_X", "?")), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
rows.sort()
t0 = rows[0][0]
with open(os.path.join(out, "dispatches.txt"), "w") as fh:
    prev_end = None
    for s, n, g, d in rows:
        short = n.replace("(anonymous namespace)::", "").split("(")[0][-70:]
        fh.write("%10.1f us  %8.1f us  grid %-9s %s\n" % ((s - t0) / 1e3, d, g, short))
print(open(os.path.join(out, "dispatches.txt")).read().count("\n"), "dispatches")
PY
grep "$FLT" $OUT/dispatches.txt | tail -40
rm -rf $OUT/t
