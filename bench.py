#!/usr/bin/env python3
"""bench.py -- frames/s of ORB extract + match on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic frames that are already
resident in HBM: ORBextractor::operator() on every frame of the batch (8-level pyramid,
FAST, quadtree, IC angle, blur, rBRIEF) followed by the brute-force Hamming
ORBmatcher::SearchByBoW of every frame against its predecessor in the batch (one
vocabulary node = all features, i.e. N1 x N2 256-bit distances + ratio test + rotation
histogram).  Workload = BASELINE.json configs[1]: 256 synthetic 640x480 frames, 1000
features, 8 levels (TUM1.yaml parameters of the reference).

Multi-GPU: one process per GPU (torchrun), each rank owns its own batch (independent
frames, no data-path collective), weak scaling; RCCL is used only for the barrier, the MAX
of the per-rank times and the all-gather of per-rank statistics.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and
`cpu_baseline` objects.  The oracle (oracle/) is only used for the cpu_baseline leg.
"""
import argparse
import importlib
import json
import os
import sys
import threading
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def level_pixels(W, H, nlevels=8, scale=1.2):
    px = []
    s = np.float32(1.0)
    for l in range(nlevels):
        inv = np.float32(1.0) / s
        px.append(int(np.rint(np.float32(W) * inv)) * int(np.rint(np.float32(H) * inv)))
        s = np.float32(s * np.float64(np.float32(scale)))
    return px


def algorithmic_bytes(W, H, K, nlevels=8):
    """stage-wise minimum traffic per frame (bytes), per stage."""
    px = level_pixels(W, H, nlevels)
    P, P0, PL = sum(px), px[0], px[-1]
    return {
        "pyramid": (P - PL) + (P - P0),        # read every level but the last, write every level but the first
        "fast_cells": P,                        # FAST + cell NMS fused: read every pyramid pixel once
        "octree": 0,
        "orient": 749 * K,                      # 749-pixel disc per keypoint
        "blur": 2 * P,                          # read + write every level
        "describe": 512 * K + 32 * K + 28 * K,  # 512 samples, 32-byte descriptor, 28-byte keypoint
        "match": 32 * (K + K) + 8 * K,          # both descriptor sets once + result per query
    }


# stage of the HIP-event timing -> kernel name in the rocprofv3 summaries under profiles/
STAGE_KERNEL = {"pyramid": "k_resize (7 launches)", "fast_cells": "k_fast_cells", "octree": "k_octree", "orient": "k_orient", "blur": "k_blur",
                "describe": "k_describe", "match_distances": "k_bow_topk", "match_replay": "k_bow_greedy"}


def pmc_traffic(stage, frames_per_launch):
    """HBM bytes per launch of the stage's kernel from the committed PMC passes (FETCH_SIZE / WRITE_SIZE,
    separate rocprofv3 --pmc runs of this same command, tools/run_profiles.sh + tools/summarize_profile.py,
    FETCH_SIZE doubled as the MI355X guide prescribes for gfx950).  None when no pass is committed for
    this launch size."""
    p = ROOT / "profiles" / "latest_hbm_traffic.json"
    if not p.exists():
        return None
    try:
        t = json.loads(p.read_text())
    except ValueError:
        return None
    if t.get("_frames_per_launch") != frames_per_launch:
        return None
    name = STAGE_KERNEL.get(stage, "").split(" ")[0]
    for k, v in t.items():
        if isinstance(v, dict) and k.split("<")[0] == name:
            return int(v["hbm_bytes_per_launch"] * (7 if stage == "pyramid" else 1))
    return None


def cpu_baseline(orbx, W, H, nf, seconds_budget=12.0):
    """Reference ORBextractor (oracle/_ref = unmodified source + cvshim) + restated matcher on
    the host cores of this box, one extractor instance per thread (instances are not
    re-entrant, reference include/ORBextractor.h:161)."""
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib
    orc = oracle_lib.Oracle()
    kind = "reference" if orc.ref is not None else "port"
    ncores = os.cpu_count() or 1
    nthreads = min(ncores, 32)
    per_thread = 6
    frames = [orbx.synth_frame(9000 + i, W, H) for i in range(per_thread + 1)]
    # calibrate on one frame so the sample stays within the budget
    ext0 = orc.reference(nf) if orc.ref is not None else orc.restatement(nf)
    t0 = time.perf_counter()
    ext0.extract(frames[0])
    one = time.perf_counter() - t0
    per_thread = int(max(2, min(200, seconds_budget / max(one * 1.3, 1e-3))))
    frames = orbx.synth_sequence(9000, per_thread + 1, W, H)
    done = [0] * nthreads

    def work(t):
        ext = orc.reference(nf) if orc.ref is not None else orc.restatement(nf)
        prev = None
        for i in range(per_thread + 1):
            k, d = ext.extract(frames[i])
            ks = np.zeros(len(k), orbx.KEYPOINT_DTYPE)
            ks["angle"] = k[:, 3]
            if prev is not None:
                oracle_lib.search_by_bow(orc, 0, prev[0], prev[1], ks, d, 0.7, True)
                done[t] += 1
            prev = (ks, d)

    th = [threading.Thread(target=work, args=(t,)) for t in range(nthreads)]
    t0 = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    dt = time.perf_counter() - t0
    total = sum(done)
    return {"value": round(total / dt, 2), "unit": "frames/s", "cores": nthreads, "kind": kind,
            "sample": "%d threads x %d frames %dx%d/%d feat, extract (%s) + brute-force SearchByBoW (restated), %.1f s"
                      % (nthreads, per_thread, W, H, nf, "oracle/_ref: unmodified reference ORBextractor.cc on cvshim" if kind == "reference" else "restatement", dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--nfeatures", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-match", action="store_true", help="extract only (BASELINE config 2)")
    ap.add_argument("--split", action="store_true", help="cut every batch into --streams parts (one launch per part) instead of "
                    "alternating whole batches over the handle pairs")
    ap.add_argument("--streams", type=int, default=3, help="extractor/matcher handle pairs (HIP stream pairs) that take the batches in turn, so that "
                    "the latency-bound stages of one batch overlap the VALU-bound stages of the others (3 measured best: 2 -> 180k, 3 -> 187k, 4 -> 174k)")
    a = ap.parse_args()

    import torch
    assert torch.cuda.is_available(), "bench.py needs a GPU: liborbx has no CPU fallback"
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev_t = torch.device("cuda", local)
    orbx = importlib.import_module("self_commit_orb-slam2_amd")
    grp = orbx.distributed.Group(device=dev_t)       # nccl (= RCCL) when WORLD_SIZE > 1
    rank, world = grp.rank, grp.world

    W, H, B, nf = a.width, a.height, a.batch, a.nfeatures
    NS = max(1, min(a.streams, B))
    if a.split:
        while B % NS:
            NS -= 1
        Bs = B // NS                                # frames per launch: the batch is cut into NS parts, one per handle pair
    else:
        Bs = B                                      # every launch covers the whole batch; consecutive steps alternate handle pairs
    exts = [orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=Bs, device=local) for _ in range(NS)]
    mts = [None if a.no_match else orbx.ORBmatcher(0.7, True, max_features=exts[0].capacity, max_pairs=Bs, device=local) for _ in range(NS)]
    ext, mt = exts[0], mts[0]
    # independent frames per rank: seeds offset by rank<<32; every 16th frame low texture
    # scenes of 16 views translating 3x1 px per view, so consecutive frames really match
    frames = orbx.synth_sequence(grp.seed_base() + 1, B, W, H)
    # inputs resident in HBM before the timed region (alternating handles each hold their own copy of the batch)
    devs = [e.upload(frames[k * Bs:(k + 1) * Bs] if a.split else frames) for k, e in enumerate(exts)]
    pa = np.arange(Bs, dtype=np.int32)              # frame i (as "KeyFrame") ...
    pb = (np.arange(Bs, dtype=np.int32) + 1) % Bs   # ... against frame i+1 (as "Frame"), inside its launch
    issued = [0]

    def step():
        # one pass of the hot path over one batch of B frames.  Nothing is synchronised between steps, so with
        # two handle pairs (two HIP stream pairs) the kernels of consecutive batches overlap on the GPU.
        if a.split:
            todo = list(range(NS))
        else:
            todo = [issued[0] % NS]
            issued[0] += 1
        for k in todo:
            exts[k].run_device(*devs[k])
        for k in todo:
            if mts[k] is not None:
                fs = orbx.ORBmatcher.features_of(exts[k], Bs)     # results are double buffered: ask every step
                mts[k].search_by_bow_device(fs, fs, pa, pb, mode=0, after=exts[k])

    def sync_all():
        for e in exts:
            e.sync()
        for m in mts:
            if m is not None:
                m.sync()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    sync_all()
    grp.barrier()
    sync_all()

    # HIP events on the library's own streams, one event set per call, recorded INSIDE the timed
    # region and read only after it (nothing is synchronised in between)
    for e in exts:
        e.set_profiling(True)
    for m in mts:
        if m is not None:
            m.sync()
            try:
                m.last_timing()     # reset the matcher's running average (warm-up calls)
            except orbx.OrbxError:
                pass
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync_all()
    grp.barrier()
    sync_all()
    elapsed = time.perf_counter() - t0
    # per-launch kernel time of every stage: each part's launch covers Bs frames
    stage_ms, timed = {}, []
    for k, e in enumerate(exts):
        try:
            tm = e.last_timing()[1]
        except orbx.OrbxError:            # a handle pair that no timed step used (steps < streams)
            tm = None
        e.set_profiling(False)
        if tm:
            timed.append(k)
            for kk, v in tm.items():
                stage_ms[kk] = stage_ms.get(kk, 0.0) + v
    stage_ms = {kk: v / max(1, len(timed)) for kk, v in stage_ms.items()}
    match_split = np.mean([mts[k].last_kernel_timing() for k in timed], axis=0) if (mt is not None and timed) else (0.0, 0.0)
    # the same stages with nothing else on the GPU: one handle pair, synchronised after every call (outside the timed region)
    alone_ms = {}
    if timed:
        k0 = timed[0]
        exts[k0].set_profiling(True)
        for _ in range(3):
            exts[k0].run_device(*devs[k0])
            exts[k0].sync()
        alone_ms = dict(exts[k0].last_timing()[1])
        exts[k0].set_profiling(False)
    parts = range(NS) if a.split else range(1)            # alternating handles hold the same batch: count it once
    counts = np.concatenate([exts[k].download(Bs)[2] for k in parts])
    nm_mean = 0.0
    if mt is not None:
        nm_mean = float(np.mean([mts[k].download(Bs)[2].mean() for k in parts]))
    t, frames_total, per_rank = grp.aggregate(elapsed, B * a.steps, int(counts.sum()))

    if rank == 0:
        K = float(counts.mean())
        alg = algorithmic_bytes(W, H, K)
        if mt is not None:
            stage_ms["match_distances"] = float(match_split[0])   # k_bow_order + k_bow_topk
            stage_ms["match_replay"] = float(match_split[1])      # k_bow_greedy (sequential greedy assignment, latency bound)
            alg["match_distances"] = alg.pop("match")
            alg["match_replay"] = 8 * int(K)                      # reads the candidate lists' heads, writes one result per query
        dom = max((k for k in stage_ms if alg.get(k, 0) > 0), key=lambda k: stage_ms[k])
        bytes_per_launch = alg[dom] * Bs                      # one launch of a part covers Bs frames
        achieved = bytes_per_launch / (stage_ms[dom] * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": STAGE_KERNEL.get(dom, dom), "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": pmc_traffic(dom, Bs),
                    "algorithmic_bytes_per_launch": int(bytes_per_launch), "avg_launch_ms": round(stage_ms[dom], 4),
                    "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
                    "frames_per_launch": Bs,
                    # the same kernel when the other stream pairs are idle (3 synchronised calls after the timed region)
                    "uncontended": ({"avg_launch_ms": round(alone_ms[dom], 4), "achieved": round(bytes_per_launch / (alone_ms[dom] * 1e-3) / 1e9, 2),
                                     "frac": round(bytes_per_launch / (alone_ms[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)} if alone_ms.get(dom) else None),
                    "uncontended_stage_ms": {k: round(v, 4) for k, v in alone_ms.items()},
                    "whole_path_algorithmic_GBs": round(sum(alg.values()) * Bs / (sum(stage_ms.values()) * 1e-3) / 1e9, 2)}
        out = {
            "metric": "frames/s ORB extract+match (1000 feat, 640x480)" if not a.no_match else "frames/s ORB extract (1000 feat, 640x480)",
            "value": round(frames_total / t, 1), "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(t / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "batch of %d synthetic %dx%d frames per GPU, ORB extract (%d feat, 8 levels, FAST 20/7)%s"
                                   % (B, W, H, nf, "" if a.no_match else " + brute-force Hamming SearchByBoW of consecutive frames"),
                       "batch_per_gpu": B, "streams": NS, "width": W, "height": H, "nfeatures": nf, "parallelism": "frames sharded, %d rank(s)" % world,
                       "keypoints_per_frame": round(K, 1), "matches_per_pair": round(nm_mean, 1)},
            "roofline": roofline,
        }
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(orbx, W, H, nf)
        print(json.dumps(out), flush=True)
    grp.close()


if __name__ == "__main__":
    main()
