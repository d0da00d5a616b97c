"""weight-stationary GRU forward (csrc/decoder5.hip) vs the weight-streaming lean kernel (decoder4.hip): bitwise, repeated"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deflow_amd.decoder import ConvGRUDecoder, PointSet
from deflow_amd._lib import img

dev = torch.device("cuda")
torch.manual_seed(0)
B, N, H = int(os.environ.get("B", 3)), int(os.environ.get("N", 1000)), 64
T = int(os.environ.get("T", 4))
head = ConvGRUDecoder(num_iters=T).to(dev)
before = torch.randn(B, H, H, 64, device=dev)
after = torch.randn(B, H, H, 64, device=dev)
coords = torch.zeros(B, N, 3, dtype=torch.int32, device=dev)
coords[..., 1:] = torch.randint(0, H, (B, N, 2), device=dev, dtype=torch.int32)
offs = (torch.rand(B, N, 3, device=dev) - 0.5) * 0.2
counts = torch.tensor([N - 37, 0, 1][:B] + [N] * max(0, B - 3), dtype=torch.int32, device=dev)
ps = PointSet(coords, offs, counts)
def run(ws, save):
    os.environ["DF_GRU_WS"] = "1" if ws else "0"
    flow, hs = head.run(img(before), img(after), ps, save)
    torch.cuda.synchronize()
    return flow.clone(), (hs.clone() if hs is not None else None)
def valid(f):
    return torch.cat([f[b, :int(counts[b])] for b in range(B)])
f4, h4 = run(False, True)
bad = 0
for rep in range(int(os.environ.get("REPS", 50))):
    f5, h5 = run(True, True)
    d = (valid(f5) - valid(f4)).abs().max().item()
    if d != 0:
        bad += 1
        idx = (valid(f5) - valid(f4)).abs().sum(1).nonzero().flatten()
        print(f"rep {rep}: max |flow5 - flow4| = {d:.3e} at rows {idx[:10].tolist()} ({idx.numel()} rows)")
print("weight-stationary vs streaming: differing repetitions", bad)
# where does it start?  hsave planes [T + 1][B*N][128]
f5, h5 = run(True, True)
P = B * N * 128
for it in range(T + 1):
    a, b_ = h4[it * P:(it + 1) * P].view(B, N, 128), h5[it * P:(it + 1) * P].view(B, N, 128)
    d = torch.cat([(a[b, :int(counts[b])] - b_[b, :int(counts[b])]).abs() for b in range(B)])
    cols = (d.max(0).values > 0).nonzero().flatten().tolist()
    print(f"plane {it}: max diff {d.max().item():.3e}, rows differing {(d.max(1).values > 0).sum().item()}, columns {cols[:12]}{'...' if len(cols) > 12 else ''} ({len(cols)})")
