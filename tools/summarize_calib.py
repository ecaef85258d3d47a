#!/usr/bin/env python3
"""Reads the rocprofv3 rocpd databases of tools/run_calib.sh (gpurun_out/calib/<counters>/calib_results.db) and prints, per kernel of
tools/ubench_hbm_calib and counter, the counter value per dispatch and its ratio to the known byte count."""
import json, re, sqlite3, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
CAL = Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "gpurun_out" / "calib"
known = {}
for line in (CAL / "known.txt").read_text().splitlines():
    m = re.match(r"known_bytes (\w+) (\d+)", line)
    if m:
        known[m.group(1)] = int(m.group(2))


def read_db(db):
    c = sqlite3.connect(str(db))
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    t = lambda stem: [x for x in tabs if x.startswith(stem)][0]
    sym = {r[0]: r[1] for r in c.execute("select id, kernel_name from %s" % t("rocpd_info_kernel_symbol"))}
    pmc = {r[0]: r[1] for r in c.execute("select id, name from %s" % t("rocpd_info_pmc"))}
    disp = {r[0]: sym.get(r[1], "?") for r in c.execute("select id, kernel_id from %s" % t("rocpd_kernel_dispatch"))}
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % t("rocpd_pmc_event"))]
    out = {}
    ev_col = "event_id" if "event_id" in cols else cols[1]
    # pmc_event rows reference the dispatch through event_id == kernel_dispatch.event_id
    ev2disp = {r[1]: r[0] for r in c.execute("select id, event_id from %s" % t("rocpd_kernel_dispatch"))}
    for r in c.execute("select %s, pmc_id, value from %s" % (ev_col, t("rocpd_pmc_event"))):
        d = ev2disp.get(r[0])
        if d is None:
            continue
        k = disp[d].split("(")[0]
        out.setdefault((k, pmc.get(r[1], str(r[1]))), []).append(r[2])
    return out


res = {}
for db in sorted(CAL.glob("*/calib_results.db")):
    try:
        for (k, cn), vals in read_db(db).items():
            res.setdefault(k, {})[cn] = sum(vals) / len(vals)
    except Exception as e:          # noqa: BLE001
        print("skip", db, e, file=sys.stderr)
table = {}
for k, cs in sorted(res.items()):
    kb = known.get(k)
    row = {"known_bytes": kb}
    for cn, v in sorted(cs.items()):
        row[cn] = v
    if kb:
        if "FETCH_SIZE" in cs:
            row["FETCH_SIZE_KB_x1024_over_known"] = round(cs["FETCH_SIZE"] * 1024 / kb, 4)
        if "WRITE_SIZE" in cs:
            row["WRITE_SIZE_KB_x1024_over_known"] = round(cs["WRITE_SIZE"] * 1024 / kb, 4)
    table[k] = row
print(json.dumps(table, indent=1))
