"""per-layer timing of the MFMA kernels in one training step at the bench shape (HIP events around every launch)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deflow_amd import ops
from deflow_amd.deflow import DeFlow
from deflow_amd.optim import Trainer
from deflow_amd.synth import synth_batch

dev = torch.device("cuda")
torch.manual_seed(0)
B = int(os.environ.get("B", 16))
model = DeFlow().to(dev)
tr = Trainer(model, lr=2e-4, dtype=(sys.argv[1] if len(sys.argv) > 1 else "fp32"))
batch = synth_batch(B, 80000, device=dev)
for _ in range(2):
    tr.step(batch)
torch.cuda.synchronize()
prof = ops.KernelProfiler()
ops.PROFILER = prof
tr.step(batch)
torch.cuda.synchronize()
ops.PROFILER = None
rows = {}
for name, flops, e0, e1, tag, *_ in prof.records:
    d = rows.setdefault((name, tag), [0, 0.0, 0.0])
    d[0] += 1; d[1] += flops; d[2] += e0.elapsed_time(e1)
tot = sum(v[2] for v in rows.values())
print(f"{'kernel':34s} {'layer':44s} {'n':>3s} {'ms':>8s} {'TF/s':>7s}  share")
for (name, tag), (n, fl, ms) in sorted(rows.items(), key=lambda kv: -kv[1][2]):
    print(f"{name:34s} {tag:44s} {n:3d} {ms:8.3f} {fl / ms / 1e9:7.1f}  {100 * ms / tot:5.1f}%")
print(f"total {tot:.2f} ms")
