#!/usr/bin/env python3
"""Kernel timeline of ONE single-frame drop-in call (after warm-up) from a rocprofv3 --kernel-trace database: start offset and duration of
every dispatch of the last complete frame.   rocprofv3 --kernel-trace -d /tmp/t -o t -- python tools/latency_single.py ; python tools/trace_single.py /tmp/t"""
import glob
import sqlite3
import sys

db = sorted(glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True))[0]
con = sqlite3.connect(db)
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
rows = list(con.execute("select name, start, end from kernels order by start"))
# the last k_orient_describe ends a frame; walk back to the k_resize run that starts it
names = [r[0] for r in rows]
pt = [i for i, n in enumerate(names) if "k_pyramid_tiles" in n]
if pt:      # the last frame that went through the single-frame graph (one k_pyramid_tiles node); frames run with stage timers use the per-level launches
    first = pt[-1]
    last = min(i for i, n in enumerate(names) if i > first and "k_orient_describe" in n)
    while first > 0 and "k_orient_describe" not in names[first - 1]:
        first -= 1
else:
    last = max(i for i, n in enumerate(names) if "k_orient_describe" in n)
    first = last
    while first > 0 and "k_orient_describe" not in names[first - 1]:
        first -= 1
t0 = rows[first][1]
for n, s, e in rows[first:last + 1]:
    short = n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    print("%-28s start %8.1f us  dur %7.1f us" % (short, (s - t0) / 1e3, (e - s) / 1e3))
print("frame span %.1f us (first kernel start -> last kernel end)" % ((rows[last][2] - t0) / 1e3))
