#!/usr/bin/env python
"""Dump the per-kernel summary (calls, total/avg duration, %) of a rocprofv3 --kernel-trace --stats run
(rocpd sqlite database) as CSV text for profiles/."""
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
with open(out, "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats summary (durations in microseconds), source: %s\n" % db.split("/")[-1])
    f.write("kernel,calls,total_us,avg_us,percent\n")
    for name, calls, tot, avg, pct in rows:
        f.write('"%s",%d,%.3f,%.3f,%.3f\n' % (name.replace('"', "'"), calls, tot, avg, pct))
print("wrote", out, len(rows), "kernels")
