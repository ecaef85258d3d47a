"""Multi-GPU plumbing: one process per GPU, frames sharded, no data-path collective.

Independent frames / frame pairs are the unit of work: every rank
extracts and matches its own batch.  The only communication of a run is
  * barriers around the timed region,
  * MAX over ranks of the elapsed time,
  * ONE all-gather of {frames, seconds, keypoints, launch-issue seconds} per rank (32 bytes),
  * before the timed region: an all-reduce of ones (rank count), an all-gather of every rank's device identity (64 bytes per rank:
    two ranks on one GPU are an error, not a scaling result) and a MAX all-reduce of the passes per step,
over `torch.distributed` - backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
"""
import os
import socket
import subprocess
import sys

import torch


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch(nproc, argv, env=None, timeout=None, capture=False):
    """Spawn `nproc` ranks of `argv` (a script path + its arguments) on this node, exactly as the bench contract
    launches them: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P <argv>.  Used by `python bench.py --gpus N` when it is started without a launcher, and by the
    CPU (gloo) test of the same path.  Returns the CompletedProcess."""
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL needs it on this driver
    e.setdefault("OMP_NUM_THREADS", "1")
    r = None
    for attempt in range(3):                              # the free port is found and released before torchrun binds it: retry a lost race
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(nproc)), "--master-addr", "127.0.0.1",
               "--master-port", str(free_port())] + list(argv)
        r = subprocess.run(cmd, env=e, timeout=timeout, stdout=subprocess.PIPE if capture else None, stderr=subprocess.PIPE if capture else None,
                           text=True if capture else None)
        if r.returncode == 0 or not capture or "ddress already in use" not in (r.stderr or ""):
            break
    return r


def device_identity(local):
    """What tells two GPUs apart: UUID and PCI address of HIP device `local`, plus the NUMA node its PCI function hangs on - from liborbx
    (orbx_device_identity), NOT from torch: a torch.cuda query here initialises torch's runtime before bench.py's handles create their
    streams, which moved those to other hardware queues and cost 13 % of the headline throughput."""
    import ctypes
    import importlib
    L = importlib.import_module(os.path.basename(os.path.dirname(os.path.abspath(__file__)))).load_library()      # (this file is loaded by path)
    buf = ctypes.create_string_buffer(96)
    node = ctypes.c_int(-1)
    L.orbx_device_identity.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p]
    rc = L.orbx_device_identity(int(local), buf, 96, ctypes.byref(node))
    if rc != 0:
        raise RuntimeError("orbx_device_identity(%d): %s" % (local, L.orbx_last_error().decode("utf-8", "replace")))
    ident = buf.value.decode()
    uuid, _, bus = ident.partition("@")
    return {"uuid": uuid, "pci_bus_id": bus, "numa_node": node.value}


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def bind_to_numa(node, rank, world):
    """Pin this process to the host cores next to its GPU: the cores of NUMA node `node` (read from the GPU's PCI function); ranks whose
    GPUs hang on the same node share that node's cores.  Without NUMA information (node < 0: single-socket box, container without
    /sys) the allowed cores are cut into `world` contiguous slices and rank r takes slice r.  Every rank issues ~20 kernel launches
    per millisecond from Python: a rank whose thread migrates across sockets shows up as a slow rank.
    -> {"numa_node", "cores": how many, "policy"}; never raises (no affinity support: policy "none")."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return {"numa_node": node, "cores": 0, "policy": "none"}
    want, policy = None, "none"
    if node >= 0:
        try:
            with open("/sys/devices/system/node/node%d/cpulist" % node) as f:
                want = sorted(_parse_cpulist(f.read()) & set(allowed))
            policy = "numa_node_of_gpu"
        except (OSError, ValueError):
            want = None
    if not want and world > 1 and len(allowed) >= world:
        per = len(allowed) // world
        want, policy = allowed[rank * per:(rank + 1) * per], "even_slices"
    if want:
        try:
            os.sched_setaffinity(0, want)
        except OSError:
            return {"numa_node": node, "cores": len(allowed), "policy": "none"}
        return {"numa_node": node, "cores": len(want), "policy": policy}
    return {"numa_node": node, "cores": len(allowed), "policy": "none"}


def emit(obj):
    """One JSON line on stdout in ONE write(2): ranks that share the launcher's pipe cannot interleave inside a line (writes of up to
    PIPE_BUF = 4096 bytes are atomic; longer lines - bench.py's - come from rank 0 only)."""
    import json
    sys.stdout.flush()
    os.write(1, (json.dumps(obj) + "\n").encode())


def parse_json_objects(text):
    """Every top-level JSON object in `text`, wherever the line breaks fell."""
    import json
    dec, out, i = json.JSONDecoder(), [], 0
    while True:
        i = text.find("{", i)
        if i < 0:
            return out
        try:
            obj, j = dec.raw_decode(text, i)
        except ValueError:
            i += 1
            continue
        if isinstance(obj, dict):
            out.append(obj)
        i = j


class Group:
    def __init__(self, backend=None, device=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        self.device = device if device is not None else torch.device("cpu")
        # a process group exists whenever a launcher started us - also with ONE rank (`torchrun --nproc-per-node 1`): the RCCL
        # initialisation, all-reduce, all-gather and barrier of an N-GPU run are then exercised on a 1-GPU box too
        launched = "TORCHELASTIC_RUN_ID" in os.environ or os.environ.get("ORBX_DIST_FORCE") == "1"
        if self.world > 1 or launched:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            if backend is None:
                backend = os.environ.get("ORBX_DIST_BACKEND") or ("nccl" if self.device.type == "cuda" else "gloo")
            if backend == "gloo":
                self.device = torch.device("cpu")      # gloo moves host tensors (CPU tests; a GPU box where the ranks share one device)
            kw = {}
            if backend == "nccl":
                kw["device_id"] = self.device
            dist.init_process_group(backend, rank=self.rank, world_size=self.world, **kw)
            self.dist = dist
            self.backend = backend
        else:
            self.backend = None

    def check(self, expected_world):
        """The process group really has `expected_world` ranks: world size as launched, and an all-reduce of ones over the
        backend (RCCL on GPUs) sums to it.  Returns {"backend", "world", "allreduce_ones"}; raises otherwise."""
        if self.world != int(expected_world):
            raise RuntimeError("launched with WORLD_SIZE=%d but %d ranks were asked for" % (self.world, int(expected_world)))
        ones = self.world
        if self.dist is not None:
            if self.dist.get_world_size() != self.world:
                raise RuntimeError("process group has %d ranks, expected %d" % (self.dist.get_world_size(), self.world))
            t = torch.ones(1, dtype=torch.float64, device=self.device)
            self.dist.all_reduce(t)
            ones = int(round(float(t.item())))
            if ones != self.world:
                raise RuntimeError("%s all-reduce saw %d ranks, expected %d" % (self.backend, ones, self.world))
        return {"backend": self.backend or "none", "world": self.world, "allreduce_ones": ones}

    def gather_identities(self, identity, allow_shared=False):
        """All-gather of every rank's device identity (a short string: GPU UUID / PCI address) -> list by rank.  Two ranks naming the
        same device raise (an "8-GPU" line measured on fewer GPUs must not exist) unless allow_shared (the 1-GPU plumbing mode)."""
        if self.dist is None:
            return [identity]      # (and no device tensor: anything that initialises torch's runtime before bench.py's handles exist moves their streams to
                                   # other hardware queues - measured: 220k instead of 252k frames/s)
        raw = identity.encode()[:64].ljust(64, b"\0")
        mine = torch.tensor(list(raw), dtype=torch.uint8, device=self.device)
        rows = [torch.zeros_like(mine) for _ in range(self.world)]
        self.dist.all_gather(rows, mine)
        ids = [bytes(r.tolist()).rstrip(b"\0").decode(errors="replace") for r in rows]
        if len(set(ids)) != len(ids) and not allow_shared:
            raise RuntimeError("ranks share a device: %s" % ", ".join("rank %d = %s" % (i, d) for i, d in enumerate(ids)))
        return ids

    def max_int(self, v):
        """MAX over ranks of a small integer (the passes per step every rank must agree on)."""
        if self.dist is None:
            return int(v)
        t = torch.tensor([float(v)], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return int(round(float(t.item())))

    def seed_base(self):
        """Seeds of rank r start at r<<32: every rank renders different frames."""
        return self.rank << 32

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def aggregate(self, elapsed, nframes, nkeypoints, issue_seconds=0.0, extra=()):
        """-> (max elapsed over ranks, total frames, per-rank [frames, seconds, keypoints, launch-issue seconds] rows).  The last column
        is the host time the rank spent issuing its launches inside the timed region: a rank whose issue time approaches its elapsed
        time is host bound, not GPU bound."""
        t = torch.tensor([elapsed], dtype=torch.float64, device=self.device)
        stats = torch.tensor([float(nframes), float(elapsed), float(nkeypoints), float(issue_seconds)] + [float(x) for x in extra], dtype=torch.float64,
                             device=self.device)      # (`extra`: further per-rank columns, e.g. NUMA node and bound cores; same length on every rank)
        if self.dist is None:
            return float(elapsed), float(nframes), [stats.tolist()]
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        rows = [torch.zeros_like(stats) for _ in range(self.world)]
        self.dist.all_gather(rows, stats)      # the one collective of this workload: 32 bytes per rank
        rows = [r.tolist() for r in rows]
        return float(t.item()), sum(r[0] for r in rows), rows

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None
