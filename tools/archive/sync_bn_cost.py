"""Cost of sync_bn on the bench step: run under `python -m torch.distributed.run --nproc-per-node 1` with
DF_FORCE_COLLECTIVES=1 (RCCL, 1-rank group: the collectives and the fp64 torch finalisation are all executed, the sums are
the identity) -- step time with and without sync_bn."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import deflow_amd
from deflow_amd.optim import Trainer
from deflow_amd.synth import synth_batch
dev = torch.device("cuda", 0)
batch = synth_batch(16, 80000, device=dev)
for sync in (False, True, False, True):
    torch.manual_seed(0)
    m = deflow_amd.DeFlow().to(dev).train()
    tr = Trainer(m, lr=2e-4, sync_bn=sync)
    for _ in range(2): tr.step(batch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): loss = tr.step(batch)
    torch.cuda.synchronize()
    print(f"sync_bn={sync}: {(time.perf_counter() - t0) / 5 * 1e3:.1f} ms/step, loss {float(loss):.5f}")
dist.destroy_process_group()
