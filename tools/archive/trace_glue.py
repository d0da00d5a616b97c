"""Which torch ops (not engine kernels) run in one B=1 inference forward: torch.profiler op list with counts"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deflow_amd
from deflow_amd.synth import synth_batch
dev = torch.device("cuda")
m = deflow_amd.DeFlow().to(dev).eval()
m.inference_dtype = os.environ.get("DF_INFER_DTYPE", "bf16")
b = synth_batch(1, 80000, device=dev)
with torch.no_grad():
    for _ in range(3):
        m.forward_padded(b)
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        m.forward_padded(b)
        torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="count", row_limit=30, max_name_column_width=60))
for e in prof.events():
    if e.name in ("aten::copy_", "aten::cat", "aten::zero_", "aten::fill_", "aten::mul", "aten::rsqrt", "aten::sub", "aten::stack") and e.stack:
        st = [s for s in e.stack if "deflow_amd" in s][:2]
        print(e.name, "|", " <- ".join(st))
