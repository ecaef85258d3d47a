#!/usr/bin/env python3
"""Per-kernel averages of the counters in one or more rocprofv3 --pmc result DBs.
    python tools/summarize_pmc.py gpurun_out/pmc_sq1 gpurun_out/pmc_sq2 ... [-o profiles/name.csv]"""
import csv
import sqlite3
import sys
from pathlib import Path


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    return name.split("(")[0].replace("void ", "").strip()


def main():
    args = sys.argv[1:]
    out = None
    if "-o" in args:
        i = args.index("-o")
        out = args[i + 1]
        args = args[:i] + args[i + 2:]
    data, counters = {}, []
    for d in args:
        con = sqlite3.connect(str(Path(d) / "bench_results.db"))
        q = "select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"
        for k, c, n, v in con.execute(q):
            data.setdefault(short(k), {})[c] = v
            data[short(k)]["launches"] = n
            if c not in counters:
                counters.append(c)
    rows = [["kernel", "launches"] + counters]
    for k in sorted(data):
        if k.startswith("__amd"):
            continue
        rows.append([k, data[k].get("launches", 0)] + ["%.0f" % data[k].get(c, 0) for c in counters])
    w = [max(len(str(r[i])) for r in rows) for i in range(len(rows[0]))]
    for r in rows:
        print("  ".join(str(x).rjust(w[i]) for i, x in enumerate(r)))
    if out:
        with open(out, "w", newline="") as f:
            csv.writer(f).writerows(rows)


if __name__ == "__main__":
    main()
