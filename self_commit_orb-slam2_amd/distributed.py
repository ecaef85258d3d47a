"""Multi-GPU plumbing: one process per GPU, frames sharded, no data-path collective.

Independent frames / frame pairs are the unit of work: every rank
extracts and matches its own batch.  The only communication of a run is
  * barriers around the timed region,
  * MAX over ranks of the elapsed time,
  * ONE all-gather of {frames, seconds, keypoints} per rank (24 bytes),
over `torch.distributed` - backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
"""
import os

import torch


class Group:
    def __init__(self, backend=None, device=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        self.device = device if device is not None else torch.device("cpu")
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            if backend is None:
                backend = "nccl" if self.device.type == "cuda" else "gloo"
            kw = {}
            if backend == "nccl":
                kw["device_id"] = self.device
            dist.init_process_group(backend, rank=self.rank, world_size=self.world, **kw)
            self.dist = dist

    def seed_base(self):
        """Seeds of rank r start at r<<32: every rank renders different frames."""
        return self.rank << 32

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def aggregate(self, elapsed, nframes, nkeypoints):
        """-> (max elapsed over ranks, total frames, per-rank [frames, seconds, keypoints] rows)."""
        t = torch.tensor([elapsed], dtype=torch.float64, device=self.device)
        stats = torch.tensor([float(nframes), float(elapsed), float(nkeypoints)], dtype=torch.float64, device=self.device)
        if self.dist is None:
            return float(elapsed), float(nframes), [stats.tolist()]
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        rows = [torch.zeros_like(stats) for _ in range(self.world)]
        self.dist.all_gather(rows, stats)      # the one collective of this workload: 24 bytes per rank
        rows = [r.tolist() for r in rows]
        return float(t.item()), sum(r[0] for r in rows), rows

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None
