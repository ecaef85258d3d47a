#!/usr/bin/env python3
"""Turn the rocprofv3 result databases written by tools/run_profiles.sh into small text
summaries under profiles/ (the .db files stay in gpurun_out/, which is scratch).

    python tools/summarize_profile.py gpurun_out/prof_<tag> profiles/<name>

writes <name>_kernel_stats.csv (rocprofv3 --kernel-trace --stats: calls, total/avg us, %) and
<name>_hbm_traffic.csv (per kernel: avg FETCH_SIZE / WRITE_SIZE in KB per launch from the two
separate --pmc passes, plus FETCH_SIZE doubled as the MI355X guide prescribes for wide
coalesced reads on gfx950).
"""
import csv
import sqlite3
import sys
from pathlib import Path


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    return name.split("(")[0].replace("void ", "").strip()


def kernel_stats(db):
    con = sqlite3.connect(db)
    rows = list(con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    return [(short(n), c, t, a, p) for n, c, t, a, p in rows]


def pmc(db, counter):
    con = sqlite3.connect(db)
    q = "select kernel_name, count(*), avg(value) from counters_collection where counter_name=? group by kernel_name"
    return {short(n): (c, v) for n, c, v in con.execute(q, (counter,))}


def main():
    src, dst = Path(sys.argv[1]), Path(sys.argv[2])
    dst.parent.mkdir(parents=True, exist_ok=True)
    ks = kernel_stats(src / "trace" / "bench_results.db")
    with open(str(dst) + "_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
        for r in ks:
            w.writerow([r[0], r[1], "%.3f" % r[2], "%.3f" % r[3], "%.2f" % r[4]])
    if (src / "trace1" / "bench_results.db").exists():      # the one-stream run: every kernel alone
        with open(str(dst) + "_streams1_kernel_stats.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
            for r in kernel_stats(src / "trace1" / "bench_results.db"):
                w.writerow([r[0], r[1], "%.3f" % r[2], "%.3f" % r[3], "%.2f" % r[4]])
    fe = pmc(src / "pmc_fetch" / "bench_results.db", "FETCH_SIZE") if (src / "pmc_fetch" / "bench_results.db").exists() else {}
    wr = pmc(src / "pmc_write" / "bench_results.db", "WRITE_SIZE") if (src / "pmc_write" / "bench_results.db").exists() else {}
    with open(str(dst) + "_hbm_traffic.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches", "FETCH_SIZE_KB_per_launch", "FETCH_SIZE_x2_KB (gfx950 correction)", "WRITE_SIZE_KB_per_launch"])
        for k in sorted(set(fe) | set(wr)):
            fv = fe.get(k, (0, 0.0))
            wv = wr.get(k, (0, 0.0))
            w.writerow([k, fv[0] or wv[0], "%.1f" % fv[1], "%.1f" % (2 * fv[1]), "%.1f" % wv[1]])
    # machine-readable per-kernel HBM traffic for bench.py's roofline.traffic (bytes per launch:
    # FETCH_SIZE is in KB and counts 64 B per 128-B request on gfx950 -> doubled, MI355X guide section HBM)
    import json
    traffic = {}
    for k in sorted(set(fe) | set(wr)):
        if k.startswith("__amd"):
            continue
        fv, wv = fe.get(k, (0, 0.0)), wr.get(k, (0, 0.0))
        traffic[k] = {"launches": fv[0] or wv[0], "fetch_kb": round(fv[1], 1), "write_kb": round(wv[1], 1),
                      "hbm_bytes_per_launch": int((2 * fv[1] + wv[1]) * 1024)}
    with open(str(dst) + "_hbm_traffic.json", "w") as f:
        json.dump(traffic, f, indent=1, sort_keys=True)
    for r in ks[:12]:
        print("%-28s calls %4d avg %10.1f us  %5.1f%%   fetch %10.0f KB  write %10.0f KB" %
              (r[0], r[1], r[3], r[4], fe.get(r[0], (0, 0))[1], wr.get(r[0], (0, 0))[1]))


if __name__ == "__main__":
    main()
