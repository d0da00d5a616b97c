"""bf16 inference only (for rocprofv3 --kernel-trace): argv[1] = "b16" (512^2, 80k, 4 it, B=16) or "cfg4" (1024^2, 160k, 8 it, B=1)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, deflow_amd
from deflow_amd.synth import synth_batch
dev = torch.device("cuda")
which = sys.argv[1] if len(sys.argv) > 1 else "b16"
kw, B, N, grid = (dict(), 16, 80000, 512) if which == "b16" else (dict(), 1, 80000, 512) if which == "b1" else (
    dict(grid_feature_size=[1024, 1024], point_cloud_range=[-102.4, -102.4, -3, 102.4, 102.4, 3], num_iters=8), 1, 160000, 1024)
torch.manual_seed(0)
m = deflow_amd.DeFlow(**kw).to(dev).eval()
m.inference_dtype = "bf16"
batch = synth_batch(B, N, grid_hw=(grid, grid), device=dev)
with torch.no_grad():
    for _ in range(3): m.forward_padded(batch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): m.forward_padded(batch)
    torch.cuda.synchronize()
print(which, "ms/forward", (time.perf_counter() - t0) / 10 * 1e3)
