#!/usr/bin/env python3
"""bench.py -- frames/s of ORB extract + match on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic frames that are already
resident in HBM: ORBextractor::operator() on every frame of the batch (8-level pyramid,
FAST, quadtree, IC angle, blur, rBRIEF) followed by the brute-force Hamming
ORBmatcher::SearchByBoW of every frame against its successor in the batch (one
vocabulary node = all features, i.e. N1 x N2 256-bit distances + ratio test + rotation
histogram).  Default workload = BASELINE.json configs[1]: 256 synthetic 640x480 frames, 1000
features, 8 levels (TUM1.yaml parameters of the reference).  `--workload sequence` =
configs[3]: every rank owns ONE synthetic sequence of --seq-len (512) frames, resident as
256-frame batches; a step is one pass over the whole sequence.  `--workload stereo` = configs[2]
(KITTI-shaped 1241x376 stereo pairs, 2000 features, extract L+R + the L<->R Hamming match), and
`--workload lba` = configs[4] (50-KF / 5000-point local BA window); those two print their own metric.

Multi-GPU: one process per GPU, each rank owns its own frames (seeds offset by rank<<32, no
data-path collective), weak scaling; RCCL carries the barrier, the MAX of the per-rank times,
one all-reduce that proves the rank count and ONE 24-byte all-gather of per-rank statistics.
`python bench.py --gpus N` started WITHOUT a launcher spawns the N ranks itself (torch.distributed.run,
127.0.0.1); started under torchrun (the driver's way) it checks WORLD_SIZE == N.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (+ `roofline_valu`)
and `cpu_baseline` objects.  The oracle (oracle/) is only used for the cpu_baseline legs.

The default command (`python bench.py --gpus 1`) measures the headline metric first (timed region and
workload as BASELINE.json's metric names them) and AFTER it, outside that timed region, the other
BASELINE configs on the same GPU, each with its own `roofline` and `cpu_baseline`, under `workloads`:
`extract_only` (configs[1] as written: extraction alone), `stereo_1241x376` (configs[2]) and `lba_50kf`
(configs[4]).  `--no-workloads` skips them.
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


_HANDLES = []      # device handles the workloads created: closed before the drop-in surface is timed (their streams would share hardware queues with it)


def close_handles():
    import gc
    while _HANDLES:
        h = _HANDLES.pop()
        try:
            (getattr(h, "close", None) or getattr(h, "destroy", None) or (lambda: None))()
        except Exception:      # noqa: BLE001
            pass
    gc.collect()


def valu_evidence():
    """VALU issue ceilings in G wave64-instructions/s for the chip (256 CUs x 4 SIMDs), with where they come from:
    profiles/r05_valu_issue.txt = the raw output of tools/ubench_valu.hip on the MI355X (63 opcodes - 27 in round 4 -, cycles per wave-instruction per SIMD
    from the kernel's duration, clock measured in the kernel): two classes, ~2.2 cycles (add / sub / and / or / xor / mov / v_lshrrev_b32 / f32 add, mul, fma,
    the two-operand 16-bit integer forms, compares) and ~4.1 cycles (left shifts, 32-bit and f32 min / max, bcnt, perm, alignbyte, dot2 / dot4, sad, mad, bfe,
    cndmask, three-operand and packed ops - what the byte kernels here are made of);
    profiles/r05_valu_mix.json = the static opcode mix of every kernel priced with that table (tools/valu_mix.py).  The guide's figure
    (MI355X_MICROARCH.md, "Wave scheduling": 2 cycles per wave64 instruction) is carried beside them."""
    clk, c_half, c_full, mix = 2.4, 4.0, 2.2, {}
    src = ["built-in defaults: profiles/r05_valu_mix.json / r04_valu_mix.json not readable"]
    for tag in ("r05", "r04"):
        try:
            d = json.loads((ROOT / "profiles" / (tag + "_valu_mix.json")).read_text())
            clk, c_half, c_full = float(d["clock_GHz_median"]), float(d["cycles_half_rate_class"]), float(d["cycles_full_rate_class"])
            mix = {k: float(v["cycles_per_instr_static_mix"]) for k, v in d["kernels"].items()}
            src = ["profiles/%s_valu_issue.txt (tools/ubench_valu.hip, raw)" % tag, "profiles/%s_valu_mix.json (tools/valu_mix.py)" % tag]
            break
        except Exception:
            continue
    simds = 256 * 4
    return {"clock_GHz": clk, "cycles_half_rate_class": c_half, "cycles_full_rate_class": c_full, "mix": mix, "evidence": src,
            "peak_half_rate": simds * clk / c_half, "peak_full_rate": simds * clk / c_full, "peak_guide_2cycle": simds * 2.4 / 2.0}


VALU = valu_evidence()
VALU_PEAK_GINST = VALU["peak_half_rate"]      # the class the byte / packed-16 kernels of this path are made of (measured)


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
STAGE_KERNEL = {"pyramid": "k_resize", "fast_cells": "k_fast_cells", "octree": "k_octree", "blur": "k_blur",
                "describe": "k_orient_describe", "match_distances": "k_bow_topk", "match_replay": "k_bow_greedy"}


def merge_orient(alg, *stage_dicts):
    """IC_Angle and the descriptor are ONE kernel (k_orient_describe): the extractor's "orient" event span is empty.  Its algorithmic
    bytes (the 749-pixel disc per keypoint) are counted with the descriptor stage."""
    if "orient" in alg:
        alg["describe"] = alg.get("describe", 0) + alg.pop("orient")
    for d in stage_dicts:
        if d and "orient" in d:
            d["describe"] = d.get("describe", 0.0) + d.pop("orient")


def csrc_sha():
    """Hash of the kernel sources the library was built from (csrc/ travels with the library).  The PMC summaries under profiles/
    carry the hash of the tree they were measured on (tools/run_profiles.sh writes it on the GPU box); counters of an older
    kernel are not reported as this run's traffic."""
    import hashlib
    h = hashlib.sha256()
    d = ROOT / "self_commit_orb-slam2_amd" / "csrc"
    for f in sorted(list(d.glob("*.hip")) + list(d.glob("*.h")) + list(d.glob("*.inc")) + list(d.glob("*.cc"))):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


_STALE = []


def _profile_json(name):
    p = ROOT / "profiles" / name
    if not p.exists():
        return None
    try:
        t = json.loads(p.read_text())
    except ValueError:
        return None
    if t.get("_csrc_sha") != csrc_sha():
        if name not in _STALE:
            _STALE.append(name)
            print("bench.py: profiles/%s was measured on other kernel sources (%s, built tree %s): its counters are not reported" %
                  (name, t.get("_csrc_sha"), csrc_sha()), file=sys.stderr)
        return None
    return t


def pmc_traffic(stage, frames_per_launch, name="latest_hbm_traffic.json"):
    """HBM bytes per launch of the stage's kernel(s) from the committed PMC passes (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc
    runs of this same command, tools/run_profiles.sh + tools/summarize_all.py; units and the gfx950 corrections as
    profiles/README.md states them).  None when no pass is committed for this launch size."""
    t = _profile_json(name)
    if not t or t.get("_frames_per_launch") != frames_per_launch:
        return None
    name = STAGE_KERNEL.get(stage, "")
    tot, found = 0, False
    for k, v in t.items():
        if isinstance(v, dict) and k.split("<")[0].split("(")[0] == name:
            tot += int(v["hbm_bytes_per_launch"] * v.get("launches_per_batch", 1))
            found = True
    return tot if found else None


def pmc_valu(stage, frames_per_launch):
    """VALU wave-instructions per launch of the stage's kernel(s) (SQ_INSTS_VALU pass, profiles/latest_sq_counters.json)."""
    t = _profile_json("latest_sq_counters.json")
    if not t or t.get("_frames_per_launch") != frames_per_launch:
        return None
    name = STAGE_KERNEL.get(stage, "")
    tot, found = 0, False
    for k, v in t.items():
        if isinstance(v, dict) and k.split("<")[0].split("(")[0] == name:
            tot += int(v["valu_insts_per_launch"] * v.get("launches_per_batch", 1))
            found = True
    return tot if found else None


def cpu_baseline(orbx, W, H, nf, seconds_budget=10.0):
    """Reference ORBextractor (oracle/_ref = unmodified source + cvshim) + restated matcher on the host cores of this box:
    (1) ONE thread = the reference's own monocular threading (src/Frame.cc:394), (2) one extractor instance per thread on all
    cores (instances are not re-entrant, reference include/ORBextractor.h:161)."""
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib
    orc = oracle_lib.Oracle()
    kind = "reference" if orc.ref is not None else "port"
    ncores = os.cpu_count() or 1
    nthreads = min(ncores, 32)
    mk = (lambda: orc.reference(nf)) if orc.ref is not None else (lambda: orc.restatement(nf))
    # calibrate on one frame so that both samples stay within the budget
    ext0 = mk()
    f0 = orbx.synth_frame(9000, W, H)
    t0 = time.perf_counter()
    ext0.extract(f0)
    one = time.perf_counter() - t0
    per_thread = int(max(2, min(200, seconds_budget / max(one * 1.3, 1e-3))))
    frames = orbx.synth_sequence(9000, per_thread + 1, W, H)

    def run(nthr, nfr):
        done = [0] * nthr

        def work(t):
            ext = mk()
            prev = None
            for i in range(nfr + 1):
                k, d = ext.extract(frames[i])
                ks = np.zeros(len(k), orbx.KEYPOINT_DTYPE)
                ks["angle"] = k[:, 3]
                if prev is not None:
                    oracle_lib.search_by_bow(orc, 0, prev[0], prev[1], ks, d, 0.7, True)
                    done[t] += 1
                prev = (ks, d)
        th = [threading.Thread(target=work, args=(t,)) for t in range(nthr)]
        t0 = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        dt = time.perf_counter() - t0
        return sum(done) / dt, dt
    single, dt1 = run(1, min(per_thread, max(2, int(4.0 / max(one * 1.3, 1e-3)))))
    multi, dtn = run(nthreads, per_thread)
    src = "oracle/_ref: unmodified reference ORBextractor.cc on cvshim" if kind == "reference" else "restatement"
    return {"value": round(multi, 2), "unit": "frames/s", "cores": nthreads, "kind": kind,
            "single_thread": {"value": round(single, 2), "cores": 1, "note": "the reference's own monocular threading (src/Frame.cc:394)", "seconds": round(dt1, 1)},
            "sample": "%d threads x %d frames %dx%d/%d feat, extract (%s) + brute-force SearchByBoW (restated), %.1f s" % (nthreads, per_thread, W, H, nf, src, dtn),
            "caveat": "cvshim's OpenCV primitives are scalar C++; a stock SIMD OpenCV build is faster, so this is a lower bound on stock ORB-SLAM2 CPU throughput"}


def measured_copy_bandwidth(torch, dev):
    """Device-to-device copy of 1 GiB: (read + write bytes) / time = what this box's HBM delivers to a plain streaming kernel."""
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device=dev)
    b = torch.empty(n, dtype=torch.uint8, device=dev)
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    del a, b
    return 2 * n / (ms * 1e-3) / 1e9


def resident_batches(orbx, torch, dev, ext, frames, B):
    """Frames as device-resident batches in the extractor's input layout (row stride / frame pitch as orbx_upload_frames lays them out);
    the memory is a torch tensor, shared read-only by every handle."""
    _, stride, pitch, (_, W, H) = ext.upload(frames[:1])
    out = []
    for b in range(0, len(frames), B):
        host = np.zeros((B, pitch), np.uint8)
        for i, im in enumerate(frames[b:b + B]):
            rows = host[i, :stride * H].reshape(H, stride)
            rows[:, :W] = im
        t = torch.from_numpy(host).to(dev)
        out.append((t, (ctypes_ptr(t), stride, pitch, (B, W, H))))
    return out


def ctypes_ptr(t):
    import ctypes
    return ctypes.c_void_p(t.data_ptr())


def bench_extract_match(a, orbx, torch, grp, dev_t, local, rank_info):
    rank, world = grp.rank, grp.world
    W, H, B, nf = a.width, a.height, a.batch, a.nfeatures
    NS = 1 if a.alone else max(1, a.streams)
    nbatches = max(1, a.batches_per_step) if a.workload == "batch" else max(1, a.seq_len // B)
    exts = [orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B, device=local) for _ in range(NS)]
    mts = [None if a.no_match else orbx.ORBmatcher(0.7, True, max_features=exts[0].capacity, max_pairs=B, device=local) for _ in range(NS)]
    mt = mts[0]
    # independent frames per rank: seeds offset by rank<<32; scenes of 16 views translating 3x1 px per view, so
    # consecutive frames really match; every 16th frame low texture
    frames = orbx.synth_sequence(grp.seed_base() + 1, B * nbatches, W, H)
    batches = resident_batches(orbx, torch, dev_t, exts[0], frames, B)     # inputs resident in HBM before the timed region
    pa = np.arange(B, dtype=np.int32)              # frame i (as "KeyFrame") ...
    pb = (np.arange(B, dtype=np.int32) + 1) % B    # ... against frame i+1 (as "Frame"), inside its launch
    issued = [0]
    passes = [1]

    def step():
        # one pass of the hot path over the rank's frames (N > 1: `passes` of them, see below).  Nothing is synchronised between
        # launches: consecutive batches go to alternating handle pairs (HIP stream pairs), so their kernels overlap on the GPU.
        for bi in [b for _ in range(passes[0]) for b in range(nbatches)]:
            k = issued[0] % NS
            issued[0] += 1
            exts[k].run_device(*batches[bi][1])
            if a.alone:
                exts[k].sync()       # --alone (profiling aid): nothing of another call is ever on the GPU next to a kernel
            if mts[k] is not None:
                fs = orbx.ORBmatcher.features_of(exts[k], B)     # results are double buffered: ask every step
                mts[k].search_by_bow_device(fs, fs, pa, pb, mode=0, after=exts[k])
                if a.alone:
                    mts[k].sync()

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
    if world > 1:
        # N > 1: the timed region of K steps must be long against the skew with which N processes leave a barrier (the driver's
        # --steps 20 are 0.16 s per rank at one pass): every step passes over the resident batches `passes` times so that the region
        # is >= ~1 s; all ranks use the MAX of their estimates.  Per-GPU work per step is the same at every N > 1 (weak scaling).
        tw = time.perf_counter()
        step()
        sync_all()
        est = max(time.perf_counter() - tw, 1e-4)
        passes[0] = grp.max_int(max(1, int(np.ceil(1.0 / (est * max(1, a.steps))))))
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
    issue_s = time.perf_counter() - t0      # host time spent issuing the launches (a rank where this approaches `elapsed` is host bound)
    sync_all()
    grp.barrier()
    sync_all()
    elapsed = time.perf_counter() - t0
    nbatches_timed = nbatches * passes[0]
    # per-launch kernel time of every stage inside the overlapped pipeline: each launch covers B frames
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
    # the same stages with nothing else on the GPU: one handle pair, synchronised after every call (outside the timed region).
    # This is what `rocprofv3 --kernel-trace --stats` of a --streams 1 run reports as the kernels' average durations.
    alone_ms, alone_match = {}, (0.0, 0.0)
    if timed:
        k0 = timed[0]
        exts[k0].set_profiling(True)
        if mts[k0] is not None:
            mts[k0].sync()
            try:
                mts[k0].last_timing()
            except orbx.OrbxError:
                pass
        for _ in range(3):
            exts[k0].run_device(*batches[0][1])
            exts[k0].sync()
            if mts[k0] is not None:
                fs = orbx.ORBmatcher.features_of(exts[k0], B)
                mts[k0].search_by_bow_device(fs, fs, pa, pb, mode=0, after=exts[k0])
                mts[k0].sync()
        alone_ms = dict(exts[k0].last_timing()[1])
        if mts[k0] is not None:
            mts[k0].last_timing()
            alone_match = mts[k0].last_kernel_timing()
        exts[k0].set_profiling(False)
    # BASELINE configs[1] as written (extraction alone), measured after the headline's timed region with the same handles and inputs
    extract_only = None
    if a.workloads and world == 1 and not a.no_match:
        def xstep():
            for bi in range(nbatches):
                k = issued[0] % NS
                issued[0] += 1
                exts[k].run_device(*batches[bi][1])
        for _ in range(2):
            xstep()
        sync_all()
        tx = time.perf_counter()
        for _ in range(a.steps):
            xstep()
        sync_all()
        tx = time.perf_counter() - tx
        extract_only = {"metric": "frames/s ORB extract (1000 feat, 640x480)", "value": round(B * nbatches * a.steps / tx, 1), "unit": "frames/s",
                        "ms_per_step": round(tx / a.steps * 1e3, 4), "steps": a.steps, "config": {"workload": "BASELINE config 2: the same %d resident batches of %d frames, "
                        "extraction only (no matcher), %d stream(s)" % (nbatches, B, NS)}}
    # the same step with the candidate lists of the matcher on the matrix cores (ORBX_MATCH_MFMA=1, read per call by liborbx): a measured
    # alternative to the popcount kernel that BASELINE's north_star prescribes - bit-identical lists - reported beside the headline, never AS it
    match_mfma = None
    if a.workloads and world == 1 and not a.no_match:
        os.environ["ORBX_MATCH_MFMA"] = "1"
        try:
            for _ in range(2):
                step()
            sync_all()
            tm_ = time.perf_counter()
            for _ in range(a.steps):
                step()
            sync_all()
            tm_ = time.perf_counter() - tm_
            mk = timed[0] if timed else 0
            mts[mk].sync()
            try:
                mts[mk].last_timing()
            except orbx.OrbxError:
                pass
            fs = orbx.ORBmatcher.features_of(exts[mk], B)
            for _ in range(3):
                mts[mk].search_by_bow_device(fs, fs, pa, pb, mode=0)
                mts[mk].sync()
            mts[mk].last_timing()
            match_mfma = {"frames_per_s": round(B * nbatches * passes[0] * a.steps / tm_, 1), "match_distances_ms_alone": round(float(mts[mk].last_kernel_timing()[0]), 4),
                          "note": "ORBX_MATCH_MFMA=1: candidate lists from v_mfma_i32_32x32x32_i8 (k_bow_topk_mfma) instead of v_xor / v_bcnt; same matches; not the default "
                                  "(north_star: Hamming match on popcount wavefront primitives, no MFMA)"}
        finally:
            del os.environ["ORBX_MATCH_MFMA"]
    _HANDLES.extend([h for h in list(exts) + list(mts) if h is not None])
    last = (issued[0] - 1) % NS
    counts = exts[last].download(B)[2]
    status = [int(e.status()) for e in exts] if hasattr(exts[0], "status") else []
    nm_mean = float(mts[last].download(B)[2].mean()) if mt is not None else 0.0
    t, frames_total, per_rank = grp.aggregate(elapsed, B * nbatches_timed * a.steps, int(counts.sum()) * nbatches_timed * a.steps, issue_s,
                                              extra=(rank_info.get("numa_node", -1), rank_info.get("bound_cores", 0)))

    if rank != 0:
        return None
    K = float(counts.mean())
    alg = algorithmic_bytes(W, H, K)
    merge_orient(alg, stage_ms, alone_ms)
    if mt is not None:
        stage_ms["match_distances"] = float(match_split[0])   # k_bow_topk
        stage_ms["match_replay"] = float(match_split[1])      # k_bow_order + k_bow_greedy (sequential greedy assignment, latency bound)
        alone_ms["match_distances"], alone_ms["match_replay"] = float(alone_match[0]), float(alone_match[1])
        alg["match_distances"] = alg.pop("match")
        alg["match_replay"] = 8 * int(K)                      # reads the candidate lists' heads, writes one result per query
    else:
        alg.pop("match")
    # dominant kernel = the longest stage when nothing else runs (= rocprofv3's per-kernel average of a --streams 1 run; the contended
    # event spans below additionally contain the time a launch waits behind the other stream pairs' workgroups)
    ref_ms = alone_ms if alone_ms else stage_ms
    dom = max((k for k in ref_ms if alg.get(k, 0) > 0), key=lambda k: ref_ms[k])
    bytes_per_launch = alg[dom] * B
    achieved = bytes_per_launch / (ref_ms[dom] * 1e-3) / 1e9
    copy_bw = measured_copy_bandwidth(torch, dev_t)
    whole_bytes = sum(alg.values()) * B * nbatches_timed
    whole_gbs = whole_bytes / (t / a.steps) / 1e9
    roofline = {"bound": "hbm", "kernel": STAGE_KERNEL.get(dom, dom), "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": pmc_traffic(dom, B),
                "algorithmic_bytes_per_launch": int(bytes_per_launch), "avg_launch_ms": round(ref_ms[dom], 4), "frames_per_launch": B,
                "timing": "HIP events on the library's stream, launches synchronised (no other stream active)",
                "contended": {"avg_launch_ms": round(stage_ms.get(dom, 0.0), 4),
                              "achieved": round(bytes_per_launch / (max(stage_ms.get(dom, 0.0), 1e-9) * 1e-3) / 1e9, 2),
                              "note": "event span of the same launch inside the %d-stream pipeline (includes queueing behind the other streams)" % NS},
                "dominant_by_contended_span": (lambda d2: {"kernel": STAGE_KERNEL.get(d2, d2) + (" (7 launches)" if d2 == "pyramid" else ""),
                                                          "span_ms": round(stage_ms[d2], 4), "achieved": round(alg[d2] * B / (stage_ms[d2] * 1e-3) / 1e9, 2),
                                                          "frac": round(alg[d2] * B / (stage_ms[d2] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                                          "note": "the stage with the longest event span inside the pipeline of this command (what a rocprofv3 trace of this "
                                                                  "command ranks first by total kernel time); for the seven dependent k_resize launches the span is mostly queueing"})(
                    max((k for k in stage_ms if alg.get(k, 0) > 0), key=lambda k: stage_ms[k])),
                "rocprof": {"alone": "profiles/*_alone_kernel_stats.csv = rocprofv3 --kernel-trace --stats of `bench.py --alone` (every launch synchronised; average durations = stage_ms_alone)",
                            "this_command": "profiles/*_kernel_stats.csv of this command (durations stretched by the concurrency of the three streams)"},
                "stage_ms_alone": {k: round(v, 4) for k, v in alone_ms.items()},
                "stage_ms_contended": {k: round(v, 4) for k, v in stage_ms.items()},
                "stage_frac_of_hbm_peak_alone": {k: round(alg[k] * B / (v * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) for k, v in alone_ms.items() if alg.get(k, 0) > 0 and v > 0},
                "measured_copy_GBs": round(copy_bw, 1), "frac_of_measured_copy": round(achieved / copy_bw, 5),
                # the two longest kernels of the path (the detector and the matcher's distance kernel) are within a few per cent of each other: whichever
                # is `kernel` above on this box, both are priced here (the matcher's is a popcount kernel: its bytes are 2 x 32 KB of descriptors per
                # frame pair, its bound is VALU issue - roofline_valu -, its HBM fraction says nothing about it)
                "longest_kernels_alone": {STAGE_KERNEL.get(k, k): {"avg_launch_ms": round(ref_ms[k], 4), "achieved": round(alg[k] * B / (ref_ms[k] * 1e-3) / 1e9, 2),
                                                                   "frac": round(alg[k] * B / (ref_ms[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
                                          for k in sorted((k for k in ref_ms if alg.get(k, 0) > 0), key=lambda k: -ref_ms[k])[:3]},
                "whole_path": {"algorithmic_bytes_per_frame": int(sum(alg.values())), "achieved": round(whole_gbs, 2), "frac": round(whole_gbs / HBM_PEAK_GBS, 5),
                               "note": "all stages' algorithmic bytes / wall time per step (the driver-visible rate)"}}
    # the bound the path actually runs into: VALU issue (integer byte arithmetic), from the committed SQ_INSTS_VALU pass
    valu = {k: pmc_valu(k, B) for k in alg}
    roofline_valu = None
    if any(v for v in valu.values()):
        tot = sum(v for v in valu.values() if v)      # (alg has an "octree" key with 0 bytes: the quadtree's instructions are in the sum)
        rate = tot * nbatches_timed / (t / a.steps) / 1e9
        dv = valu.get(dom)
        # the same instructions priced kernel by kernel with the static opcode mix: seconds per batch if every SIMD issued without a bubble
        def kcyc(stage):
            kn = STAGE_KERNEL.get(stage, stage).split("<")[0].split(" ")[0]
            return VALU["mix"].get(kn, VALU["cycles_half_rate_class"])
        t_mix = sum(v * kcyc(k) for k, v in valu.items() if v) / (256 * 4 * VALU["clock_GHz"] * 1e9)
        peak_mix = tot / t_mix / 1e9 if t_mix > 0 else VALU_PEAK_GINST
        roofline_valu = {"bound": "valu_issue", "unit": "G wave-instr/s", "peak": round(VALU_PEAK_GINST, 1),
                         "achieved": round(rate, 2), "frac": round(rate / VALU_PEAK_GINST, 4),
                         "peak_basis": "measured: %.2f cycles per wave64 instruction per SIMD for the half-rate class (left shifts, 32-bit min / max, bcnt, perm, alignbyte, dot2 / dot4, sad, "
                                       "cndmask, packed-16), %.2f for add / sub / and / or / xor / mov / f32 add / fma / two-operand 16-bit, at %.2f GHz" % (VALU["cycles_half_rate_class"], VALU["cycles_full_rate_class"], VALU["clock_GHz"]),
                         "evidence": VALU["evidence"],
                         "peak_static_mix": round(peak_mix, 1), "frac_static_mix": round(rate / peak_mix, 4),
                         "peak_guide_2cycle": round(VALU["peak_guide_2cycle"], 1), "frac_guide_2cycle": round(rate / VALU["peak_guide_2cycle"], 4),
                         "cycles_per_instr_static_mix": {STAGE_KERNEL.get(k, k): round(kcyc(k), 2) for k, v in valu.items() if v},
                         "valu_insts_per_batch": int(tot), "source": "profiles/latest_sq_counters.json (SQ_INSTS_VALU pass of this command)",
                         "dominant_kernel": ({"kernel": STAGE_KERNEL.get(dom, dom), "valu_insts_per_launch": int(dv),
                                              "frac_alone": round(dv / (ref_ms[dom] * 1e-3) / 1e9 / VALU_PEAK_GINST, 4)} if dv else None)}
    metric = "frames/s ORB extract+match (1000 feat, 640x480)" if not a.no_match else "frames/s ORB extract (1000 feat, 640x480)"
    wl = ("batch of %d synthetic %dx%d frames per GPU (%d distinct resident batches per step, one launch set per batch)" % (B, W, H, nbatches)) if a.workload == "batch" else \
         ("one synthetic sequence of %d %dx%d frames per GPU (BASELINE config 4), resident as %d batches of %d" % (B * nbatches, W, H, nbatches, B))
    out = {
        "metric": metric, "value": round(frames_total / t, 1), "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(t / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": wl + ", ORB extract (%d feat, 8 levels, FAST 20/7)%s" % (nf, "" if a.no_match else " + brute-force Hamming SearchByBoW of consecutive frames"),
                   "batch_per_gpu": B, "batches_per_step": nbatches, "passes_per_step": passes[0], "streams": NS, "width": W, "height": H, "nfeatures": nf,
                   "parallelism": "frames sharded, %d rank(s), no data-path collective" % world,
                   "keypoints_per_frame": round(K, 1), "matches_per_pair": round(nm_mean, 1)},
        "ranks": dict({k: v for k, v in rank_info.items() if k not in ("numa_node", "bound_cores")},
                      per_rank=[{"frames": r[0], "seconds": round(r[1], 6), "keypoints": r[2], "frames_per_s": round(r[0] / r[1], 1),
                                 "launch_issue_seconds": round(r[3], 6), "issue_frac": round(r[3] / r[1], 3), "numa_node": int(r[4]), "bound_cores": int(r[5]),
                                 "device": (rank_info.get("devices") or [None] * len(per_rank))[i]} for i, r in enumerate(per_rank)]),
        "roofline": roofline,
    }
    if roofline_valu:
        out["roofline_valu"] = roofline_valu
    if status:
        out["config"]["device_status"] = status
    if extract_only:
        # roofline of the extraction alone: same dominant kernel, same launches (the matcher is the only thing removed)
        xb = sum(v for k, v in alg.items() if not k.startswith("match")) * B * nbatches
        extract_only["roofline"] = {"bound": "hbm", "kernel": roofline["kernel"], "achieved": roofline["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": roofline["frac"],
                                    "traffic": roofline["traffic"], "whole_path_GBs": round(xb / (extract_only["ms_per_step"] * 1e-3) / 1e9, 1),
                                    "note": "same launches as the headline line minus the matcher: the dominant kernel and its duration alone are the headline's"}
        out["workloads"] = {"extract_only": extract_only}
        if match_mfma:
            out["workloads"]["extract_match_mfma_opt_in"] = match_mfma
    if _STALE:
        out["roofline"]["traffic_note"] = "profiles/%s measured on other kernel sources: counters withheld" % ", ".join(_STALE)
    return out


def stereo_cpu_baseline(orbx, W, H, nf, bf, seconds_budget=8.0):
    """The reference's own stereo Frame constructor (oracle/_ref/liborbslam.so: src/Frame.cc:118-199 = two ORBextractor threads, src/Frame.cc:159-167, +
    Frame::ComputeStereoMatches, :1026-1420, all compiled unmodified on cvshim) on the same synthetic pairs: one constructor at a time (what System::TrackStereo
    does), and one per two host threads on all cores."""
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib
    lib = oracle_lib.slam_lib()
    if lib is None:
        return None
    fx, fy, cx, cy = 718.856, 718.856, 607.1928, 185.2157      # KITTI00-02.yaml:9-12
    pairs = [(orbx.synth_frame(7000 + i, W, H), orbx.synth_frame(7000 + i, W, H, orbx.SYNTH_STEREO_RIGHT)) for i in range(8)]
    t0 = time.perf_counter()
    oracle_lib.ref_stereo_frame(pairs[0][0], pairs[0][1], nf, fx, fy, cx, cy, bf, lib=lib)
    one = time.perf_counter() - t0
    ncores = os.cpu_count() or 1
    nthr = max(1, min(ncores, 32) // 2)                          # every constructor runs two extractor threads of its own

    def run(nt, per):
        def work(t):
            for i in range(per):
                oracle_lib.ref_stereo_frame(*pairs[(t + i) % len(pairs)], nf, fx, fy, cx, cy, bf, lib=lib)
        th = [threading.Thread(target=work, args=(t,)) for t in range(nt)]
        t0 = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        dt = time.perf_counter() - t0
        return nt * per / dt, dt
    # (oracle/refslam_wrap.cc gives every extractor thread the reference spawns a fresh 64 MiB bump chunk out of 1024 - the quadtree's pointer
    # tie-break, DESIGN section 3 - and never reuses one: the sample stays well below 500 constructors per process)
    n1 = int(max(2, min(16, 0.3 * seconds_budget / max(one, 1e-3))))
    single, dt1 = run(1, n1)
    nn = int(max(2, min(320 // nthr, 0.7 * seconds_budget / max(one * 1.5, 1e-3))))
    multi, dtn = run(nthr, nn)
    return {"value": round(multi, 2), "unit": "pairs/s", "cores": 2 * nthr, "kind": "reference",
            "single": {"value": round(single, 2), "cores": 2, "note": "one stereo Frame constructor at a time = the reference's own threading (two extractor threads, src/Frame.cc:159-167)",
                       "seconds": round(dt1, 1)},
            "sample": "%d concurrent stereo Frame constructors x %d pairs %dx%d/%d feat (oracle/_ref/liborbslam.so: unmodified ORBextractor.cc + Frame.cc on cvshim), %.1f s" %
                      (nthr, nn, W, H, nf, dtn),
            "caveat": "cvshim's OpenCV primitives are scalar C++; a stock SIMD OpenCV build is faster, so this is a lower bound on stock ORB-SLAM2 CPU throughput"}


def bench_stereo(a, orbx, torch, grp, dev_t, local, rank_info):
    """BASELINE configs[2]: KITTI-shaped stereo pairs 1241x376, 2000 features: extract left + right, then the L<->R Hamming match of
    Frame::ComputeStereoMatches (complete: row bands, SAD refinement, median cut) on the device."""
    W, H, nf = 1241, 376, 2000
    B = a.batch if a.batch != 256 else 64                  # pairs per batch
    NS = 1 if a.alone else max(1, a.streams)              # handle pairs (= HIP stream pairs) that take the batches in turn, as in the default workload
    exts = [orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=2 * B, device=local) for _ in range(NS)]   # KITTI00-02.yaml:41-50
    mts = [orbx.ORBmatcher(0.7, True, max_features=exts[0].capacity, max_pairs=B, device=local) for _ in range(NS)]
    ext, mt = exts[0], mts[0]
    _HANDLES.extend(list(exts) + list(mts))
    seeds = [grp.seed_base() + 100 + i for i in range(B)]
    frames = [orbx.synth_frame(s, W, H) for s in seeds] + [orbx.synth_frame(s, W, H, orbx.SYNTH_STEREO_RIGHT) for s in seeds]
    devs = [resident_batches(orbx, torch, dev_t, e, frames, 2 * B)[0] for e in exts]
    fl, fr = np.arange(B, dtype=np.int32), np.arange(B, 2 * B, dtype=np.int32)
    bf = 386.1448                                         # KITTI00-02.yaml:25

    def step(i, alone=False):
        k = i % NS
        exts[k].run_device(*devs[k][1])                   # left and right images of the batch in one launch set
        if alone:
            exts[k].sync()
        mts[k].compute_stereo_matches_device(exts[k], exts[k], fl, fr, bf, 0.0)
        if alone:
            mts[k].sync()

    def sync_all():
        for e, m in zip(exts, mts):
            e.sync(); m.sync()
        torch.cuda.synchronize()
    for i in range(a.warmup + NS):
        step(i, a.alone)
    sync_all(); grp.barrier(); sync_all()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i, a.alone)
    sync_all(); grp.barrier(); sync_all()
    elapsed = time.perf_counter() - t0
    # ONE handle pair, every step synchronised: the latency chain of a 64-pair batch (what a caller that waits for each batch's result gets; this is
    # the "17k pairs/s" figure of tools/bench_configs.py, against the pipelined `value` of three handle pairs)
    nchain = max(3, a.steps // 2)
    sync_all()
    tcs = time.perf_counter()
    for i in range(nchain):
        step(0, False)
        mts[0].sync()
    chain = (time.perf_counter() - tcs) / nchain
    # every stage with nothing else on the GPU (HIP events on the library's streams, calls synchronised)
    ext.set_profiling(True)
    mt.sync()
    try:
        mt.last_timing()
    except orbx.OrbxError:
        pass
    for _ in range(3):
        step(0, True)
    alone = dict(ext.last_timing()[1])
    alone["stereo_match"] = float(mt.last_timing())
    ext.set_profiling(False)
    u, z = mt.download_stereo(B)
    cnt = ext.download(2 * B)[2]
    t, total, per_rank = grp.aggregate(elapsed, B * a.steps, int(cnt.sum()))
    if grp.rank != 0:
        return None
    K = float(cnt.mean())
    nmatch = float((u >= 0).sum(1).mean())
    alg = algorithmic_bytes(W, H, K)
    alg.pop("match")
    merge_orient(alg, alone)
    per_image = sum(alg.values())
    # Frame::ComputeStereoMatches per pair (both descriptor sets once + one result per left keypoint) + the SAD windows of the refined
    # matches (11x11 left patch, 11 rows x 21 columns right strip, src/Frame.cc:1248-1292) + mvuRight / mvDepth
    alg_match = 32 * (K + K) + 8 * K + (121 + 231) * nmatch + 8 * K
    stage_alg = {k: v * 2 * B for k, v in alg.items()}
    stage_alg["stereo_match"] = alg_match * B
    dom = max((k for k in alone if stage_alg.get(k, 0) > 0), key=lambda k: alone[k])
    achieved = stage_alg[dom] / (alone[dom] * 1e-3) / 1e9
    kern = dict(STAGE_KERNEL, stereo_match="k_stereo_rows + k_stereo_full + k_stereo_cut")
    whole = (per_image * 2 + alg_match) * total / t / 1e9
    roof = {"bound": "hbm", "kernel": kern.get(dom, dom), "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": pmc_traffic(dom, 2 * B, "latest_stereo_hbm_traffic.json"), "algorithmic_bytes_per_launch": int(stage_alg[dom]), "avg_launch_ms": round(alone[dom], 4),
            "images_per_launch": 2 * B, "algorithmic_bytes_per_image": int(per_image), "algorithmic_bytes_per_pair": int(per_image * 2 + alg_match),
            "timing": "HIP events on the library's streams, calls synchronised (no other stream active)",
            "stage_ms_alone": {k: round(v, 4) for k, v in alone.items()},
            "stage_frac_of_hbm_peak_alone": {k: round(stage_alg[k] / (v * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) for k, v in alone.items() if stage_alg.get(k, 0) > 0 and v > 0},
            "whole_path": {"achieved": round(whole, 2), "frac": round(whole / HBM_PEAK_GBS, 5)},
            "rocprof": "profiles/*_stereo_kernel_stats.csv (this workload, pipelined) and *_stereo_alone_kernel_stats.csv (--alone: average durations = stage_ms_alone)"}
    out = {"metric": "stereo pairs/s ORB extract L+R + ComputeStereoMatches (2000 feat, 1241x376)", "value": round(total / t, 1), "unit": "pairs/s",
           "images_per_s": round(2 * total / t, 1),
           "n_gpus": grp.world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(t / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "u8", "data": "synthetic",
           "config": {"workload": "BASELINE config 3: batch of %d KITTI-shaped stereo pairs, extract left + right (2000 feat) + L<->R match on the device" % B, "streams": NS,
                      "keypoints_per_image": round(K, 1), "stereo_matches_per_pair": round(nmatch, 1)},
           "single_handle_synchronised": {"pairs_per_s": round(B / chain, 1), "ms_per_batch": round(chain * 1e3, 3), "sum_of_stages_alone_ms": round(sum(alone.values()), 3),
                                          "note": "one handle pair, the host waits for every batch: the latency chain (stage sum + launch gaps + the host's issue time); "
                                                  "`value` overlaps %d such chains on %d stream pairs" % (NS, NS)},
           "ranks": dict(rank_info, per_rank=[{"pairs": r[0], "seconds": round(r[1], 6)} for r in per_rank]), "roofline": roof}
    if not a.no_cpu_baseline and grp.world == 1:
        cb = stereo_cpu_baseline(orbx, W, H, nf, bf)
        if cb:
            out["cpu_baseline"] = cb
    return out


def bench_lba(a, orbx, torch, grp, dev_t, local, rank_info):
    """BASELINE configs[4]: Optimizer::LocalBundleAdjustment on the synthetic 50-KF / 5000-point window (replicas only: a window does not
    shard).  `value` = windows/s of ONE window at a time (what LocalMapping does: latency); `concurrent` = the same with --streams
    independent windows in flight on as many handles / host threads (a window occupies a few CUs at a time: the LM loop is a chain of
    small dependent kernels), i.e. what a batch of independent maps gets."""
    w = orbx.lba_synth.make_window(K=50, P=5000, seed=12345 + grp.rank)
    opt = orbx.Optimizer(max_keyframes=64, max_points=6000, max_edges=80000, device=local)
    for _ in range(a.warmup):
        opt.LocalBundleAdjustment(w)
    grp.barrier()
    t0 = time.perf_counter()
    kern = []
    for _ in range(a.steps):
        opt.LocalBundleAdjustment(w)
        kern.append(opt.last_timing()[0])
    grp.barrier()
    elapsed = time.perf_counter() - t0
    ms, flops = opt.last_timing()
    # ---- S windows in flight (outside the timed region of `value`)
    S = max(1, a.streams)
    opts = [opt] + [orbx.Optimizer(max_keyframes=64, max_points=6000, max_edges=80000, device=local) for _ in range(S - 1)]
    ws = [w] + [orbx.lba_synth.make_window(K=50, P=5000, seed=777 + 13 * i + grp.rank) for i in range(S - 1)]
    _HANDLES.extend(opts)
    for o, x in zip(opts[1:], ws[1:]):
        o.LocalBundleAdjustment(x)

    def work(i):
        for _ in range(a.steps):
            opts[i].LocalBundleAdjustment(ws[i])
    th = [threading.Thread(target=work, args=(i,)) for i in range(S)]
    tc = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    conc = S * a.steps / (time.perf_counter() - tc)
    t, total, per_rank = grp.aggregate(elapsed, a.steps, w["E"])
    if grp.rank != 0:
        return None
    cpu = None
    if not a.no_cpu_baseline and grp.world == 1:
        # (1) the REFERENCE: the vendored g2o (Thirdparty/g2o, compiled unmodified on oracle/eigenshim) driven with the graph and the two-stage
        #     schedule of Optimizer::LocalBundleAdjustment (src/Optimizer.cc:629-997; oracle/refslam_wrap.cc: orbslam_g2o_ba), one core - g2o's
        #     solver is single threaded in ORB-SLAM2;  (2) the restatement of the same LM (oracle/lba_oracle.cc, pinned to (1) to 1e-15)
        sys.path.insert(0, str(ROOT / "tests"))
        import oracle_lib
        orc = oracle_lib.Oracle()

        def timed(fn, budget, nmax):
            fn()
            n, tc0 = 0, time.perf_counter()
            while n < nmax and time.perf_counter() - tc0 < budget:
                fn()
                n += 1
            return n, time.perf_counter() - tc0
        n, dt = timed(lambda: oracle_lib.local_bundle_adjustment(orc, w), 6.0, 40)
        port = {"value": round(n / dt, 2), "unit": "windows/s", "cores": 1, "kind": "port", "sample": "%d windows of the same problem, %.1f s" % (n, dt),
                "note": "oracle/lba_oracle.cc: restatement of the two-stage LM with a dense Cholesky of the reduced system"}
        cpu = port
        if oracle_lib.slam_lib() is not None:
            n2, dt2 = timed(lambda: oracle_lib.g2o_ba_f64(w), 8.0, 20)
            cpu = {"value": round(n2 / dt2, 2), "unit": "windows/s", "cores": 1, "kind": "reference", "sample": "%d windows of the same problem, %.1f s" % (n2, dt2),
                   "note": "oracle/_ref/liborbslam.so: the reference's vendored g2o (BlockSolver_6_3 + LinearSolverEigen + Levenberg, sparse LDLT) compiled unmodified, "
                           "same graph and schedule as Optimizer::LocalBundleAdjustment (src/Optimizer.cc:629-997)",
                   "caveat": "built on oracle/eigenshim, an unoptimised stand-in for Eigen (eager loops, no SIMD kernels): slower than a stock Eigen build, a lower bound "
                             "on the reference's CPU speed",
                   "restatement": port}
    gflops = flops / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    launches = None      # kernel launches per window, from the committed rocprofv3 kernel trace of this workload (k_unpack runs once per window)
    try:
        import csv
        f = sorted((ROOT / "profiles").glob("r*_lba_kernel_stats.csv"))[-1]
        rows = list(csv.DictReader(open(f)))
        wins = [int(r["calls"]) for r in rows if r["kernel"].startswith("k_unpack")]
        if wins and wins[0] > 0:
            chol = [r for r in rows if r["kernel"].startswith("k_chol_step")]
            launches = {"per_window": round(sum(int(r["calls"]) for r in rows if r["kernel"].startswith("k_")) / wins[0], 1),
                        "k_chol_step_per_window": round(sum(int(r["calls"]) for r in chol) / wins[0], 1),
                        "k_chol_step_percent_of_kernel_time": round(sum(float(r["percent"]) for r in chol), 1), "source": "profiles/" + f.name}
    except Exception:
        launches = None
    FP64_PEAK_TF = 256 * 4 * 16 * 2 * 2.4e9 / 1e12      # 256 CUs x 4 SIMDs x 16 FP64 lanes x FMA x 2.4 GHz = 78.6 TFLOP/s (vector = matrix rate on MI355X)
    roof = {"bound": "latency (fp64)", "kernel": "whole LM loop (%s x k_chol_step = %s %% of the kernel time, profiles/*_lba_kernel_stats.csv)"
                                       % (("%g" % launches["k_chol_step_per_window"], "%g" % launches["k_chol_step_percent_of_kernel_time"]) if launches else ("72", "~49")),
            "achieved": round(gflops / 1e3, 4),
            "peak": round(FP64_PEAK_TF, 1), "unit": "TFLOP/s", "frac": round(gflops / 1e3 / FP64_PEAK_TF, 5), "traffic": None,
            "note": "a 50-keyframe window is ~0.4 GFLOP in ~200 dependent launches: bound by dependent FP64 latencies (pivot chains, ~16 cycles per "
                    "dependent instruction) and launch boundaries, neither by FP64 throughput nor by HBM; `concurrent` shows what independent windows recover"}
    out = {"metric": "local BA windows/s (50 KF / 5000 points)", "value": round(total / t, 2), "unit": "windows/s", "n_gpus": grp.world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(t / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic", "config": {"workload": "BASELINE config 5: LocalBundleAdjustment, %d keyframes, %d points, %d edges; replicas only" % (w["K"], w["P"], w["E"]),
                                            "kernel_ms": round(float(np.mean(kern)), 4), "fp64_gflops": round(flops / (ms * 1e-3) / 1e9, 2) if ms > 0 else None,
                                            "launches": launches, "concurrent": {"windows_in_flight": S, "windows_per_s": round(conc, 1)}},
            "ranks": dict(rank_info, per_rank=[{"windows": r[0], "seconds": round(r[1], 6)} for r in per_rank]), "roofline": roof}
    if cpu:
        out["cpu_baseline"] = cpu
    return out


def host_io_batch(orbx, a, local, seconds=1.0):
    """orbx_extract_batch as the C ABI declares it for host callers: host image pointers in, host keypoint / descriptor arrays out - upload, the
    batch's launch set and the download of the results all inside the timed region (PCIe both ways).  Three forms: the synchronous call (chunks of
    64 frames through the two-slot pipeline inside the call), orbx_extract_batch_begin / _end with two batches in flight (pageable frames), and the
    same with the frames in pinned memory (read in place, no staging copy)."""
    W, H, B, nf = a.width, a.height, min(a.batch, 256), a.nfeatures
    ext = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B, device=local)
    frames = orbx.synth_sequence(991, B, W, H)
    res = ext.extract_batch(frames)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        kps, desc, counts = ext.extract_batch(frames, out=res)
        n += 1
    dt = time.perf_counter() - t0
    cap = ext.capacity
    out = {"frames_per_s": round(n * B / dt, 1), "batch": B, "ms_per_batch": round(dt / n * 1e3, 3), "keypoints_per_frame": round(float(counts.mean()), 1),
           "note": "host pointers in (pageable), host arrays out, synchronous: %d MB up and %d MB of result arrays down per call; the result arrays are reused" % ((B * W * H) >> 20, (B * cap * 60) >> 20)}

    def piped(fr):
        r2 = [res, tuple(np.zeros_like(x) for x in res)]
        ext.extract_batch_begin(fr)
        m, t1 = 0, time.perf_counter()
        while time.perf_counter() - t1 < seconds:
            ext.extract_batch_begin(fr)
            ext.extract_batch_end(out=r2[m & 1])
            m += 1
        ext.extract_batch_end(out=r2[m & 1])
        d1 = time.perf_counter() - t1
        return {"frames_per_s": round((m + 1) * B / d1, 1), "ms_per_batch": round(d1 / (m + 1) * 1e3, 3)}
    try:
        out["pipelined"] = dict(piped(frames), note="orbx_extract_batch_begin / _end, two batches in flight, pageable frames staged on the copy pool")
        import torch
        pin = torch.empty((B, H, W), dtype=torch.uint8).pin_memory()
        hp = pin.numpy()
        for i in range(B):
            hp[i] = frames[i]
        out["pipelined_pinned"] = dict(piped([hp[i] for i in range(B)]), note="the same with the frames in pinned memory: read in place by the DMA engines")
    except Exception as e:      # noqa: BLE001
        out["pipelined_error"] = "%s: %s" % (type(e).__name__, e)
    ext.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--nfeatures", type=int, default=1000)
    ap.add_argument("--workload", choices=["batch", "sequence", "stereo", "lba"], default="batch",
                    help="batch = BASELINE configs[1] (+ match; the metric); sequence = configs[3]; stereo = configs[2]; lba = configs[4]")
    ap.add_argument("--seq-len", type=int, default=512, help="frames per sequence of --workload sequence")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-match", action="store_true", help="extract only (BASELINE config 2)")
    ap.add_argument("--batches-per-step", type=int, default=8, help="distinct resident batches of --batch frames a step of the default workload passes over "
                    "(8 x 256 x 640x480 = 630 MB of input, more than the 256 MiB Infinity Cache; the timed region of 20 steps is ~0.2 s)")
    ap.add_argument("--no-workloads", dest="workloads", action="store_false", help="headline metric only: skip the extract-only / stereo / LBA legs measured after it")
    ap.add_argument("--alone", action="store_true", help="profiling aid: ONE handle pair and a synchronisation after every call, so that a rocprofv3 kernel trace of this "
                    "command holds every kernel's duration with nothing else on the GPU (profiles/*_alone_kernel_stats.csv)")
    ap.add_argument("--streams", type=int, default=3, help="extractor/matcher handle pairs (HIP stream pairs) that take the batches in turn, so that "
                    "the latency-bound stages of one batch overlap the VALU-bound stages of the others")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started without a launcher: spawn the N ranks exactly as the contract launches them, one per GPU, and pass the result through
        dist_mod = importlib.import_module("self_commit_orb-slam2_amd.distributed")
        r = dist_mod.launch(a.gpus, [str(Path(__file__).resolve())] + sys.argv[1:])
        sys.exit(r.returncode)

    # The surface the reference's callers can actually call (ORBextractor::operator(), the stereo Frame constructor through the C++ shim) is
    # timed FIRST, in a process of its own, before this process creates a HIP context: the reference's callers are a C++ program with nothing
    # else on the device.  (Which hardware queues the library's streams land on depends on what else created streams before: the stereo
    # constructor - two launch sets, the match and the frame-finish kernel side by side - moves by +-25 us with it, the one-thread call does not.)
    dropin_digest = None
    if a.workloads and a.workload == "batch" and int(os.environ.get("WORLD_SIZE", "1")) == 1 and a.gpus == 1:
        try:
            import subprocess
            r = subprocess.run([sys.executable, str(ROOT / "tools" / "latency_shim.py"), "--quick"], capture_output=True, text=True, timeout=300)
            lines = [l for l in r.stdout.splitlines() if l.startswith('{"digest"')]
            if not lines:
                raise RuntimeError("no digest line (rc %d): %s" % (r.returncode, r.stderr[-300:]))
            dropin_digest = json.loads(lines[-1])["digest"]
            dropin_digest["process"] = "own process (tools/latency_shim.py --quick) on the same GPU, before bench.py's own context exists"
        except Exception as e:      # noqa: BLE001
            dropin_digest = {"error": "%s: %s" % (type(e).__name__, e)}

    import torch
    assert torch.cuda.is_available(), "bench.py needs a GPU: liborbx has no CPU fallback"
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("ORBX_BENCH_SHARE_GPU") == "1":      # plumbing check on a 1-GPU box: every rank on device 0, ORBX_DIST_BACKEND=gloo (RCCL refuses two ranks per GPU)
        local = local % torch.cuda.device_count()
    assert local < torch.cuda.device_count(), "rank %d has no GPU (%d visible)" % (local, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev_t = torch.device("cuda", local)
    orbx = importlib.import_module("self_commit_orb-slam2_amd")
    grp = orbx.distributed.Group(device=dev_t)       # nccl (= RCCL) when WORLD_SIZE > 1
    rank_info = grp.check(a.gpus)                    # WORLD_SIZE == --gpus, and an RCCL all-reduce of ones sees every rank
    # every rank names its GPU (UUID @ PCI address): N ranks must sit on N DISTINCT devices, otherwise the run fails here instead of
    # producing an "N-GPU" line measured on fewer GPUs (ORBX_BENCH_SHARE_GPU=1: the 1-GPU plumbing mode, marked in the line)
    ident = orbx.distributed.device_identity(local)
    shared = os.environ.get("ORBX_BENCH_SHARE_GPU") == "1"
    rank_info["devices"] = grp.gather_identities("%s@%s" % (ident["uuid"], ident["pci_bus_id"]), allow_shared=shared)
    rank_info["distinct_devices"] = len(set(rank_info["devices"]))
    if shared:
        rank_info["shared_gpu_plumbing_mode"] = True
    # N > 1: each rank on the host cores of its GPU's NUMA node (restored before rank 0's cpu_baseline, which uses every core)
    affinity0 = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
    bind = orbx.distributed.bind_to_numa(ident["numa_node"], grp.rank, grp.world) if grp.world > 1 else {"numa_node": ident["numa_node"], "cores": len(affinity0 or ()), "policy": "none (single rank)"}
    rank_info["numa_node"], rank_info["bound_cores"], rank_info["affinity_policy"] = bind["numa_node"], bind["cores"], bind["policy"]

    fn = {"batch": bench_extract_match, "sequence": bench_extract_match, "stereo": bench_stereo, "lba": bench_lba}[a.workload]
    if a.workload != "batch" or grp.world > 1:
        a.workloads = False                          # the other configs are N = 1 lines riding on the default command
    out = fn(a, orbx, torch, grp, dev_t, local, rank_info)
    if grp.rank == 0 and a.workloads:
        # BASELINE configs[2] and configs[4] on the same GPU, after (outside) the headline's timed region; same step / warm-up counts
        import copy
        for key, f2, st in (("stereo_1241x376", bench_stereo, max(a.steps, 12)), ("lba_50kf", bench_lba, max(a.steps, 10))):
            a2 = copy.copy(a)
            a2.steps, a2.warmup, a2.batch = st, max(a.warmup, 2), 256
            try:
                r = f2(a2, orbx, torch, grp, dev_t, local, rank_info)
                for k in ("n_gpus", "higher_is_better", "scaling", "vs_baseline", "data", "ranks", "warmup"):
                    r.pop(k, None)
                out.setdefault("workloads", {})[key] = r
            except Exception as e:                   # a failing side leg must not take the headline line with it
                out.setdefault("workloads", {})[key] = {"error": "%s: %s" % (type(e).__name__, e)}
    if grp.rank == 0 and a.workloads and out is not None:
        # Everything the side legs measured, once more as a COMPACT digest inside `config` (the driver's parser keeps `config` and `roofline`
        # verbatim and drops unknown top-level keys): BASELINE configs 2 / 3 / 5, the batch entry point with host pointers in and host arrays
        # out (PCIe both ways), and the surface the reference can actually call - ORBextractor::operator() and the stereo Frame constructor
        # through the C++ shim, timed by C++ loops (tools/latency_shim.py).
        w = out.get("workloads", {})
        st, lb, xo = w.get("stereo_1241x376", {}) or {}, w.get("lba_50kf", {}) or {}, w.get("extract_only", {}) or {}
        mm = w.get("extract_match_mfma_opt_in", {}) or {}
        dig = {"extract_only": {"frames_per_s": xo.get("value")},
               "extract_match_mfma_opt_in": {"frames_per_s": mm.get("frames_per_s"), "match_distances_ms_alone": mm.get("match_distances_ms_alone")},
               "stereo": {"pairs_per_s": st.get("value"), "frac_hbm": (st.get("roofline") or {}).get("frac"),
                          "cpu_ref_pairs_per_s": (st.get("cpu_baseline") or {}).get("value"), "cpu_ref_one_ctor_pairs_per_s": ((st.get("cpu_baseline") or {}).get("single") or {}).get("value")},
               "lba": {"ms_per_window": lb.get("ms_per_step"), "windows_per_s_3_in_flight": ((lb.get("config") or {}).get("concurrent") or {}).get("windows_per_s"),
                       "frac_fp64": (lb.get("roofline") or {}).get("frac"), "cpu_ref_g2o_windows_per_s": (lb.get("cpu_baseline") or {}).get("value"),
                       "cpu_port_windows_per_s": ((lb.get("cpu_baseline") or {}).get("restatement") or {}).get("value")}}
        try:
            dig["host_io_batch"] = host_io_batch(orbx, a, local)
        except Exception as e:
            dig["host_io_batch"] = {"error": "%s: %s" % (type(e).__name__, e)}
        dig["dropin"] = dropin_digest if dropin_digest is not None else {"error": "not measured"}
        # BASELINE config 4 at N = 1: one resident 512-frame sequence, one pass per step (extract + match), same handles / streams as the headline
        try:
            import copy
            a4 = copy.copy(a)
            a4.workload, a4.workloads, a4.no_cpu_baseline, a4.steps, a4.warmup = "sequence", False, True, max(4, min(a.steps, 10)), 2
            r4 = bench_extract_match(a4, orbx, torch, grp, dev_t, local, rank_info)
            dig["sequence_512"] = {"frames_per_s": r4.get("value"), "ms_per_pass": r4.get("ms_per_step")}
        except Exception as e:      # noqa: BLE001
            dig["sequence_512"] = {"error": "%s: %s" % (type(e).__name__, e)}
        # The driver's parser keeps the SCALAR keys of `config` only (nested dicts are dropped, strings cut): every number of the digest once more
        # as a flat scalar, and the nested form as the LAST key of the whole line (so that it is what a `tail` of the output shows).
        dp, hb = dig["dropin"], dig["host_io_batch"]
        # ... in order of importance, ahead of the workload's own parameters, 24 scalars in all (the parser keeps about that many): the side workloads'
        # headline numbers, the reference's own metric for the drop-in (per-frame tracking time of the call chain src/Tracking.cc makes, beside the
        # all-reference library on the same frames), the call latencies, the host-pointer batch rates.  Everything else stays in the nested digest.
        flat = {"extract_only_fps": dig["extract_only"]["frames_per_s"], "stereo_pairs_per_s": dig["stereo"]["pairs_per_s"],
                "lba_ms_per_window": dig["lba"]["ms_per_window"], "lba_launches_per_window": (((lb.get("config") or {}).get("launches") or {}).get("per_window")),
                "sequence_512_fps": dig["sequence_512"].get("frames_per_s"),
                "dropin_track_ms_median": dp.get("track_ms_median"), "dropin_track_ms_mean": dp.get("track_ms_mean"), "ref_track_ms_median": dp.get("ref_track_ms_median"),
                "dropin_us_1thread": dp.get("us_1thread"), "dropin_fps_16threads": dp.get("fps_16threads"),
                "stereo_ctor_us_median": dp.get("stereo_frame_ctor_median_us"),
                "host_io_pinned_fps": (hb.get("pipelined_pinned") or {}).get("frames_per_s"), "host_io_pipelined_fps": (hb.get("pipelined") or {}).get("frames_per_s")}
        cfg = out["config"]
        out["config"] = dict([("workload", cfg.get("workload"))] + list(flat.items()) + [(k, v) for k, v in cfg.items() if k != "workload"])
        digest_tail = dig
    else:
        digest_tail = None
    grp.close()                                      # (ranks > 0 are done; rank 0 alone times the host baseline below)
    if affinity0 is not None:
        try:
            os.sched_setaffinity(0, affinity0)
        except OSError:
            pass
    if grp.rank == 0:
        if not a.no_cpu_baseline and a.workload in ("batch", "sequence"):
            # N > 1: a short sample, so that the line of every N carries the host number of the same run
            out["cpu_baseline"] = cpu_baseline(orbx, a.width, a.height, a.nfeatures, seconds_budget=10.0 if grp.world == 1 else 4.0)
        if digest_tail is not None:
            out["workloads_digest"] = digest_tail     # LAST key of the line: the nested form of config's flat dropin_* / stereo_* / lba_* / host_io_* scalars
        orbx.distributed.emit(out)                   # one write(2) for the whole line


if __name__ == "__main__":
    main()
