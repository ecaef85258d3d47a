"""world_size-2 run of the multi-GPU plumbing on CPU (gloo): barriers, MAX of the per-rank
times, the single all-gather of per-rank statistics, per-rank seeds.  On the GPU the same code
runs over nccl (= RCCL), one process per GPU, launched by torchrun as the bench contract says."""
import os
import socket
import subprocess
import sys
import textwrap
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent

WORKER = textwrap.dedent("""
    import importlib, json, sys
    sys.path.insert(0, %r)
    orbx = importlib.import_module("self_commit_orb-slam2_amd")
    grp = orbx.distributed.Group(backend="gloo")
    # what bench.py does per rank, with the GPU work replaced by known numbers
    frames = orbx.synth_sequence(grp.seed_base() + 1, 2, 320, 240)
    checksum = int(sum(int(f.astype("int64").sum()) for f in frames))
    grp.barrier()
    elapsed = 0.5 + 0.25 * grp.rank          # the last rank is the slow one
    ids = grp.gather_identities("GPU-%%04d@0000:%%02x:00.0" %% (grp.rank, grp.rank))
    passes = grp.max_int(1 + grp.rank)       # every rank must leave with the MAX
    t, total, rows = grp.aggregate(elapsed, 100 * (grp.rank + 1), 1000 + grp.rank, 0.125 * (grp.rank + 1), extra=(grp.rank %% 2, 16))
    print(json.dumps({"rank": grp.rank, "world": grp.world, "t": t, "total": total, "rows": rows, "checksum": checksum, "ids": ids, "passes": passes}), flush=True)
    grp.close()
""") % str(ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_gloo():
    import json
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=240)
        assert p.returncode == 0, e[-2000:]
        outs.append(json.loads([l for l in o.splitlines() if l.startswith("{")][-1]))
    outs.sort(key=lambda d: d["rank"])
    for d in outs:
        assert d["world"] == 2
        assert abs(d["t"] - 0.75) < 1e-12                 # MAX over ranks
        assert d["total"] == 300                          # frames of all ranks
        assert d["rows"] == [[100.0, 0.5, 1000.0, 0.125, 0.0, 16.0], [200.0, 0.75, 1001.0, 0.25, 1.0, 16.0]]
        assert d["ids"] == ["GPU-0000@0000:00:00.0", "GPU-0001@0000:01:00.0"] and d["passes"] == 2
    assert outs[0]["checksum"] != outs[1]["checksum"]     # different frames per rank (seed = rank << 32)
    # whole-job throughput as bench.py computes it
    assert abs(outs[0]["total"] / outs[0]["t"] - 400.0) < 1e-9


def _run_world(n, worker=WORKER, extra_env=None):
    import json
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, "-c", worker], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    res = []
    for p in procs:
        o, e = p.communicate(timeout=300)
        res.append((p.returncode, o, e))
    return res


def test_four_ranks_with_a_slow_one_gloo():
    """World 4: the job's time is the SLOWEST rank's (rank 3 here), the total is every rank's frames, the per-rank rows show who was slow
    and how much of its time went into issuing launches; identities are gathered in rank order."""
    import json
    res = _run_world(4)
    outs = []
    for rc, o, e in res:
        assert rc == 0, e[-2000:]
        outs.append(json.loads([l for l in o.splitlines() if l.startswith("{")][-1]))
    outs.sort(key=lambda d: d["rank"])
    for d in outs:
        assert d["world"] == 4 and abs(d["t"] - 1.25) < 1e-12 and d["total"] == 1000 and d["passes"] == 4
        assert [r[1] for r in d["rows"]] == [0.5, 0.75, 1.0, 1.25] and [r[3] for r in d["rows"]] == [0.125, 0.25, 0.375, 0.5]
        assert len(set(d["ids"])) == 4
    assert len({d["checksum"] for d in outs}) == 4
    # efficiency as the driver would compute it from the per-N values: the slow rank costs the whole job
    assert abs(outs[0]["total"] / outs[0]["t"] - 800.0) < 1e-9


def test_two_ranks_on_one_device_are_refused():
    """An "N-GPU" line measured on fewer GPUs must not exist: identical device identities fail the run in every rank."""
    worker = WORKER.replace('"GPU-%04d@0000:%02x:00.0" % (grp.rank, grp.rank)', '"GPU-0000@0000:00:00.0"')
    assert worker != WORKER
    for rc, o, e in _run_world(2, worker):
        assert rc != 0 and "ranks share a device" in e


def test_numa_binding_falls_back_to_even_slices(orbx):
    import multiprocessing as mp
    if not hasattr(os, "sched_getaffinity") or len(os.sched_getaffinity(0)) < 2:
        import pytest
        pytest.skip("needs two allowed cores")

    def child(q, rank):
        q.put((orbx.distributed.bind_to_numa(-1, rank, 2), sorted(os.sched_getaffinity(0))))
    ctx = mp.get_context("fork")
    q = ctx.Queue()
    sets = []
    for r in range(2):
        p = ctx.Process(target=child, args=(q, r))
        p.start()
        info, cpus = q.get(timeout=60)
        p.join()
        assert info["policy"] == "even_slices" and info["cores"] == len(cpus)
        sets.append(set(cpus))
    assert not (sets[0] & sets[1]), "two ranks were pinned to overlapping cores"


def test_two_ranks_through_the_bench_spawn_path(orbx):
    """`python bench.py --gpus N` without a launcher calls distributed.launch (torch.distributed.run, 127.0.0.1, one rank per
    GPU); the same call here with the gloo worker: both ranks come up, the rank count is checked by an all-reduce."""
    import json
    r = orbx.distributed.launch(2, [str(ROOT / "tests" / "dist_worker.py"), "2"], timeout=300, capture=True)
    assert r.returncode == 0, r.stderr[-3000:]
    outs = sorted((d for d in orbx.distributed.parse_json_objects(r.stdout) if "rank" in d), key=lambda d: d["rank"])
    assert [d["rank"] for d in outs] == [0, 1]
    for d in outs:
        assert d["info"] == {"backend": "gloo", "world": 2, "allreduce_ones": 2}
        assert abs(d["t"] - 0.75) < 1e-12 and d["total"] == 300
    assert outs[0]["checksum"] != outs[1]["checksum"]
    # a rank count that does not match what was asked for is an error, not a silent n_gpus: 1
    bad = orbx.distributed.launch(2, [str(ROOT / "tests" / "dist_worker.py"), "3"], timeout=300, capture=True)
    assert bad.returncode != 0 and "ranks were asked for" in bad.stderr


def test_bench_refuses_a_mismatched_world(orbx):
    grp = orbx.distributed.Group()
    import pytest
    with pytest.raises(RuntimeError):
        grp.check(8)
    assert grp.check(1) == {"backend": "none", "world": 1, "allreduce_ones": 1}


def test_single_rank_needs_no_process_group(orbx):
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    grp = orbx.distributed.Group()
    t, total, rows = grp.aggregate(1.5, 30, 7)
    assert (t, total, rows) == (1.5, 30.0, [[30.0, 1.5, 7.0, 0.0]])
    assert grp.gather_identities("GPU-x") == ["GPU-x"] and grp.max_int(3) == 3
    grp.close()
