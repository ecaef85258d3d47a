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
    elapsed = 0.5 + 0.25 * grp.rank          # rank 1 is the slow one
    t, total, rows = grp.aggregate(elapsed, 100 * (grp.rank + 1), 1000 + grp.rank)
    print(json.dumps({"rank": grp.rank, "world": grp.world, "t": t, "total": total, "rows": rows, "checksum": checksum}), flush=True)
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
        assert d["rows"] == [[100.0, 0.5, 1000.0], [200.0, 0.75, 1001.0]]
    assert outs[0]["checksum"] != outs[1]["checksum"]     # different frames per rank (seed = rank << 32)
    # whole-job throughput as bench.py computes it
    assert abs(outs[0]["total"] / outs[0]["t"] - 400.0) < 1e-9


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
    assert (t, total, rows) == (1.5, 30.0, [[30.0, 1.5, 7.0]])
    grp.close()
