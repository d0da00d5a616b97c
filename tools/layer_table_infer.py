"""per-layer timing of one inference forward (HIP events around every launch): B from argv[1] (default 1), dtype argv[2]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deflow_amd
from deflow_amd import ops
from deflow_amd.synth import synth_batch

dev = torch.device("cuda")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
torch.manual_seed(0)
m = deflow_amd.DeFlow().to(dev).eval()
if len(sys.argv) > 2:
    m.inference_dtype = sys.argv[2]
batch = synth_batch(B, 80000, device=dev)
with torch.no_grad():
    for _ in range(3):
        m.forward_padded(batch)
    torch.cuda.synchronize()
    prof = ops.KernelProfiler()
    ops.PROFILER = prof
    m.forward_padded(batch)
    torch.cuda.synchronize()
    ops.PROFILER = None
tot = 0.0
print(f"{'kernel':30s} {'layer':46s} {'us':>8s} {'TF/s':>7s}")
for name, flops, e0, e1, tag, *_ in prof.records:
    ms = e0.elapsed_time(e1)
    tot += ms
    print(f"{name:30s} {tag:46s} {ms * 1e3:8.1f} {flops / max(ms, 1e-6) / 1e9:7.1f}")
print(f"total (profiled launches) {tot:.3f} ms")
