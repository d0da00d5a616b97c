import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deflow_amd.decoder import ConvGRUDecoder, PointSet
from deflow_amd._lib import img
dev = torch.device("cuda"); torch.manual_seed(0)
B, N, H, T = 1, 64, 64, 1
head = ConvGRUDecoder(num_iters=T).to(dev)
before = torch.randn(B, H, H, 64, device=dev); after = torch.randn(B, H, H, 64, device=dev)
coords = torch.zeros(B, N, 3, dtype=torch.int32, device=dev); coords[..., 1:] = torch.randint(0, H, (B, N, 2), device=dev, dtype=torch.int32)
offs = (torch.rand(B, N, 3, device=dev) - 0.5) * 0.2
counts = torch.tensor([N], dtype=torch.int32, device=dev)
ps = PointSet(coords, offs, counts)
def run(ws):
    os.environ["DF_GRU_WS"] = "1" if ws else "0"
    flow, hs = head.run(img(before), img(after), ps, True)
    torch.cuda.synchronize()
    return hs.view(T + 1, N, 128).clone()
h4 = run(False)
outs = [run(True) for _ in range(4)]
for k, h5 in enumerate(outs):
    d = (h5[1] - h4[1]).abs()
    print(f"run {k}: max {d.max().item():.3e}; by wave(16-col block) max:", [f"{d[:, 16*w:16*w+16].max().item():.1e}" for w in range(8)])
    print("        by point tile max:", [f"{d[16*t:16*t+16].max().item():.1e}" for t in range(4)], " fwd5 self-diff vs run0:", (h5[1] - outs[0][1]).abs().max().item())
