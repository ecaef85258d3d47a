#!/usr/bin/env python3
"""Average duration per kernel from a rocprofv3 --kernel-trace result database:  python tools/kernel_avg.py <dir> [top]"""
import glob
import sqlite3
import sys

dbs = sorted(glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True))
if not dbs:
    print("no *_results.db under", sys.argv[1])
    sys.exit(0)
con = sqlite3.connect(dbs[0])
agg = {}
for name, s, e in con.execute("select name, start, end from kernels"):
    short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    a = agg.setdefault(short, [0, 0.0])
    a[0] += 1
    a[1] += (e - s) / 1e3
tot = sum(v[1] for v in agg.values())
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[: int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print("%-34s calls %6d  avg %8.2f us  total %9.1f us  %5.1f %%" % (k, c, t / c, t, 100 * t / tot))
