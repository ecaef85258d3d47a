#!/usr/bin/env python3
"""profiles/latest_hbm_traffic.json and profiles/latest_sq_counters.json, the two files bench.py reads for roofline.traffic and
roofline_valu, from the summaries of tools/summarize_profile.py (FETCH_SIZE / WRITE_SIZE passes) and tools/summarize_pmc.py
(SQ_INSTS_VALU pass) of `python bench.py --steps 10 --warmup 2 --no-cpu-baseline`.
    python tools/make_latest_profiles.py profiles/<tag>_hbm_traffic.json profiles/<tag>_sq_counters.csv 256"""
import csv, json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
traffic, sq, B = Path(sys.argv[1]), Path(sys.argv[2]), int(sys.argv[3])
LAUNCHES = {"k_resize": 7}       # launches of that kernel per batch (one per pyramid level above the first)
t = json.loads(traffic.read_text())
out = {"_frames_per_launch": B, "_source": "%s (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of `python bench.py --steps 10 --warmup 2 --no-cpu-baseline`; "
       "FETCH_SIZE doubled: 128-byte fabric requests are tallied at 64 B on gfx950 for every read width, profiles/r02_hbm_counter_calibration.json)" % traffic}
for k, v in t.items():
    v = dict(v)
    v["launches_per_batch"] = LAUNCHES.get(k.split("<")[0], 1)
    out[k] = v
(ROOT / "profiles" / "latest_hbm_traffic.json").write_text(json.dumps(out, indent=1, sort_keys=True))
if sq.exists():
    o2 = {"_frames_per_launch": B, "_source": "%s (rocprofv3 --pmc SQ_INSTS_VALU ... pass of the same command)" % sq}
    for r in csv.DictReader(sq.open()):
        o2[r["kernel"]] = {"valu_insts_per_launch": int(float(r["SQ_INSTS_VALU"])), "launches_per_batch": LAUNCHES.get(r["kernel"].split("<")[0], 1),
                           "waves": int(float(r.get("SQ_WAVES", 0) or 0))}
    (ROOT / "profiles" / "latest_sq_counters.json").write_text(json.dumps(o2, indent=1, sort_keys=True))
print("ok")
