#!/usr/bin/env python3
"""Summaries of the rocprofv3 result databases of tools/run_profiles.sh, written on the GPU box.
    python tools/summarize_all.py <raw dir> <summary dir> <tag>
-> <tag>_kernel_stats.csv / _alone_kernel_stats.csv / _stereo_kernel_stats.csv / _stereo_alone_kernel_stats.csv /
   _lba_kernel_stats.csv  (rocprofv3 --kernel-trace --stats: calls, total / average duration per kernel),
   <tag>_hbm_traffic.{csv,json}, <tag>_stereo_hbm_traffic.json (FETCH_SIZE / WRITE_SIZE passes: bytes per launch; FETCH_SIZE x 2, see
   profiles/r02_hbm_counter_calibration.json), <tag>_sq_counters.csv, and the files bench.py reads: latest_hbm_traffic.json,
   latest_stereo_hbm_traffic.json, latest_sq_counters.json - each stamped with the hash of the kernel sources it was measured on."""
import csv
import json
import sqlite3
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    return name.split("(")[0].replace("void ", "").strip()


def db_of(d):
    c = sorted(Path(d).rglob("*_results.db"))
    return str(c[0]) if c else None


def kernel_stats(d, dst):
    db = db_of(d)
    if not db:
        return None
    con = sqlite3.connect(db)
    rows = [(short(n), c, t, a, p) for n, c, t, a, p in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels")]
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
        for r in rows:
            w.writerow([r[0], r[1], "%.3f" % r[2], "%.3f" % r[3], "%.2f" % r[4]])      # top_kernels durations are in microseconds
    print("==", dst)
    for r in rows[:14]:
        print("  %-28s calls %5d avg %10.2f us  %5.1f%%" % (r[0], r[1], r[3], r[4]))
    return rows


def pmc(d):
    db = db_of(d)
    if not db:
        return {}
    con = sqlite3.connect(db)
    out = {}
    for k, c, n, v in con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
        if short(k).startswith("__amd"):
            continue
        out.setdefault(short(k), {})[c] = v
        out[short(k)]["launches"] = n
    return out


LAUNCHES = {"k_resize": 7}      # launches of that kernel per batch


def traffic(raw, prefix, dst_csv, dst_json, latest, frames_per_launch, sha):
    fe, wr = pmc(raw / (prefix + "pmc_fetch")), pmc(raw / (prefix + "pmc_write"))
    if not fe and not wr:
        return
    t = {}
    rows = [["kernel", "launches", "FETCH_SIZE_KB_per_launch", "FETCH_SIZE_x2_KB (gfx950 correction)", "WRITE_SIZE_KB_per_launch"]]
    for k in sorted(set(fe) | set(wr)):
        f, w = fe.get(k, {}).get("FETCH_SIZE", 0.0), wr.get(k, {}).get("WRITE_SIZE", 0.0)
        n = fe.get(k, {}).get("launches", 0) or wr.get(k, {}).get("launches", 0)
        rows.append([k, n, "%.1f" % f, "%.1f" % (2 * f), "%.1f" % w])
        t[k] = {"launches": n, "fetch_kb": round(f, 1), "write_kb": round(w, 1), "hbm_bytes_per_launch": int((2 * f + w) * 1024),
                "launches_per_batch": LAUNCHES.get(k.split("<")[0], 1)}
    if dst_csv:
        with open(dst_csv, "w", newline="") as f:
            csv.writer(f).writerows(rows)
    meta = {"_frames_per_launch": frames_per_launch, "_csrc_sha": sha,
            "_source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (separate runs) of bench.py, tools/run_profiles.sh; FETCH_SIZE doubled: 128-byte fabric "
                       "requests are tallied at 64 B on gfx950 for every read width (profiles/r02_hbm_counter_calibration.json)"}
    Path(dst_json).write_text(json.dumps(dict(t, **meta), indent=1, sort_keys=True))
    Path(latest).write_text(json.dumps(dict(t, **meta), indent=1, sort_keys=True))
    print("==", dst_json)
    for k, v in t.items():
        print("  %-28s %10.1f KB fetch x2 + %10.1f KB write per launch" % (k, 2 * v["fetch_kb"], v["write_kb"]))


def main():
    raw, dst, tag = Path(sys.argv[1]), Path(sys.argv[2]), sys.argv[3]
    dst.mkdir(parents=True, exist_ok=True)
    import bench
    sha = bench.csrc_sha()
    (dst / (tag + "_csrc_sha.txt")).write_text(sha + "\n")
    for d, name in (("trace", "_kernel_stats.csv"), ("alone", "_alone_kernel_stats.csv"), ("st_trace", "_stereo_kernel_stats.csv"),
                    ("st_alone", "_stereo_alone_kernel_stats.csv"), ("lba_trace", "_lba_kernel_stats.csv")):
        if (raw / d).exists():
            kernel_stats(raw / d, dst / (tag + name))
    traffic(raw, "", dst / (tag + "_hbm_traffic.csv"), dst / (tag + "_hbm_traffic.json"), dst / "latest_hbm_traffic.json", 256, sha)
    traffic(raw, "st_", None, dst / (tag + "_stereo_hbm_traffic.json"), dst / "latest_stereo_hbm_traffic.json", 128, sha)
    sq = pmc(raw / "pmc_sq")
    sq2 = pmc(raw / "pmc_sq2")
    if sq:
        counters = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_BUSY_CYCLES"]
        rows = [["kernel", "launches"] + counters]
        latest = {"_frames_per_launch": 256, "_csrc_sha": sha, "_source": "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES pass of bench.py (tools/run_profiles.sh)"}
        for k in sorted(sq):
            v = dict(sq2.get(k, {}), **sq[k])
            rows.append([k, v.get("launches", 0)] + ["%.0f" % v.get(c, 0) for c in counters])
            latest[k] = {"valu_insts_per_launch": int(v.get("SQ_INSTS_VALU", 0)), "launches_per_batch": LAUNCHES.get(k.split("<")[0], 1), "waves": int(v.get("SQ_WAVES", 0))}
        with open(dst / (tag + "_sq_counters.csv"), "w", newline="") as f:
            csv.writer(f).writerows(rows)
        (dst / "latest_sq_counters.json").write_text(json.dumps(latest, indent=1, sort_keys=True))
        print("==", tag + "_sq_counters.csv")
        for r in rows:
            print("  " + "  ".join(str(x).rjust(14) for x in r))


if __name__ == "__main__":
    main()
