import sys, os, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch, deflow_amd
from deflow_amd.optim import Trainer
from deflow_amd.synth import synth_batch
dev = torch.device("cuda")
m = deflow_amd.DeFlow().to(dev).train()
tr = Trainer(m, lr=2e-4)
batch = synth_batch(16, 80000, device=dev)
for _ in range(3): tr.step(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): tr.step(batch)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host enqueue {t_host/10*1e3:.1f} ms/step, wall {t_all/10*1e3:.1f} ms/step")
