"""Host cost of one training step: Python enqueueing the ~410 launches (eager) vs replaying the captured program.

    python tools/host_time.py [fp32|bf16]                      one rank
    DF_FORCE_COLLECTIVES=1 python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 tools/host_time.py bf16
                                                               the data-parallel path through RCCL on a 1-rank group: the step is
                                                               captured as graph SEGMENTS with real (1-rank) all-reduces between them

"host" = wall time of the call(s) that enqueue ONE step with an empty GPU queue (a synchronize in front of every sample:
back-to-back steps block on the queue depth, which is GPU time)."""
import os
import statistics
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import deflow_amd
from deflow_amd.optim import Trainer
from deflow_amd.synth import synth_batch

dtype = sys.argv[1] if len(sys.argv) > 1 else "fp32"
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
if "RANK" in os.environ:
    import torch.distributed as dist
    dist.init_process_group("nccl", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]), device_id=dev)
m = deflow_amd.DeFlow().to(dev).train()
tr = Trainer(m, lr=2e-4, dtype=dtype)
batch = synth_batch(16, 80000, device=dev)


def sample(fn, n=7):
    hs, ws = [], []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        hs.append((time.perf_counter() - t0) * 1e3)
        torch.cuda.synchronize()
        ws.append((time.perf_counter() - t0) * 1e3)
    return statistics.median(hs), statistics.median(ws)


for _ in range(3):
    tr.step(batch)
h_e, w_e = sample(lambda: tr.step(batch))
tr.capture(batch)
for _ in range(2):
    tr.step_captured()
h_g, w_g = sample(lambda: tr.step_captured())
kinds = [o[0] for o in tr._program]
print(f"{dtype} collective={tr.collective}: eager host {h_e:.1f} ms / wall {w_e:.1f} ms per step; captured host {h_g:.2f} ms / wall {w_g:.1f} ms "
      f"({kinds.count('graph')} graph segments, {sum(len(o[1]) for o in tr._program if o[0] == 'allreduce')} all-reduce calls)")
