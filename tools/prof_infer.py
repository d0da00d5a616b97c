"""fp32 inference only (for rocprofv3 --kernel-trace): B from argv[1] (default 1), 512^2, 80k points, 4 iterations"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, deflow_amd
from deflow_amd.synth import synth_batch
dev = torch.device("cuda")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
torch.manual_seed(0)
m = deflow_amd.DeFlow().to(dev).eval()
batch = synth_batch(B, 80000, device=dev)
with torch.no_grad():
    for _ in range(3): m.forward_padded(batch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): m.forward_padded(batch)
    torch.cuda.synchronize()
print("B", B, "ms/forward", (time.perf_counter() - t0) / 10 * 1e3)
