import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time, deflow_amd
from deflow_amd.optim import Trainer
from deflow_amd.synth import synth_batch
dev = torch.device("cuda")
for B in (7, 32, 1):
    torch.manual_seed(0)
    m = deflow_amd.DeFlow().to(dev).train()
    tr = Trainer(m, lr=2e-4)
    batch = synth_batch(B, 80000, device=dev)
    for _ in range(2): loss = tr.step(batch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): loss = tr.step(batch)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f"B={B}: {dt*1e3:.1f} ms/step, {B/dt:.1f} pairs/s, loss {float(loss):.4f}, mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
    del tr, m, batch; torch.cuda.empty_cache()
