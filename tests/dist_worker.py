"""One rank of the CPU (gloo) run of the multi-GPU plumbing: what bench.py does per rank, with the GPU work replaced by known
numbers.  Launched by self_commit_orb-slam2_amd.distributed.launch (= the spawn path of `python bench.py --gpus N`)."""
import importlib
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
orbx = importlib.import_module("self_commit_orb-slam2_amd")

expected = int(sys.argv[1])
grp = orbx.distributed.Group(backend="gloo")
info = grp.check(expected)                      # WORLD_SIZE as launched + an all-reduce of ones
frames = orbx.synth_sequence(grp.seed_base() + 1, 2, 320, 240)
checksum = int(sum(int(f.astype("int64").sum()) for f in frames))
grp.barrier()
elapsed = 0.5 + 0.25 * grp.rank                 # rank 1 is the slow one
t, total, rows = grp.aggregate(elapsed, 100 * (grp.rank + 1), 1000 + grp.rank)
orbx.distributed.emit({"rank": grp.rank, "world": grp.world, "t": t, "total": total, "rows": rows, "checksum": checksum, "info": info})
grp.close()
