"""The bench's training step with Trainer(dtype=...) alone: ms per step at B=16 (A/B runs of kernel-form switches).
usage: python tools/bench_bf16_train.py [bf16|fp32] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deflow_amd
from deflow_amd.optim import Trainer
from deflow_amd.synth import synth_batch
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda")
torch.manual_seed(0)
m = deflow_amd.DeFlow().to(dev).train()
tr = Trainer(m, lr=2e-4, dtype=dtype)
b = synth_batch(16, 80000, device=dev)
for _ in range(3):
    tr.step(b)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(steps):
    loss = tr.step(b)
torch.cuda.synchronize()
ms = (time.perf_counter() - t) / steps * 1e3
env = {k: v for k, v in os.environ.items() if k.startswith("DF_")}
print(f"{dtype} train step: {ms:.2f} ms  ({16e3 / ms:.1f} pairs/s)  loss {float(loss):.4f}  {env}")
