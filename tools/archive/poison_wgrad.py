"""wgrad parity with POISONED allocator memory (NaN-filled blocks freed right before), to expose reads of unwritten
or out-of-tile data that fresh zero pages hide."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from deflow_amd import ops
from deflow_amd._lib import img, img_pair, DfImg
dev = torch.device("cuda")

def poison():
    junk = [torch.full((64 << 20,), float("nan"), device=dev) for _ in range(4)]
    torch.cuda.synchronize(); del junk

def check(name, got, want):
    got = got.double().cpu(); want = want.double()
    bad = not torch.isfinite(got).all()
    e = float((got - want).abs().max() / want.abs().max()) if not bad else float("nan")
    print(f"{name}: rel_err={e:.3e} {'NON-FINITE' if bad else ''}")

g = torch.Generator().manual_seed(0)
for (cin, cout, k, s, n, h, w) in [(64, 64, 3, 1, 2, 16, 16), (32, 64, 3, 2, 2, 16, 16), (256, 128, 1, 1, 2, 8, 16), (128, 256, 3, 1, 1, 8, 8), (64, 64, 1, 1, 2, 12, 20), (128, 64, 3, 1, 1, 24, 8)]:
    x = torch.randn(n, cin, h, w, generator=g, requires_grad=True)
    wt = (torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)).requires_grad_(True)
    y = F.conv2d(x, wt, None, stride=s, padding=k // 2)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    for rep in range(2):
        poison()
        xd = x.detach().permute(0, 2, 3, 1).contiguous().to(dev)
        gyd = gy.permute(0, 2, 3, 1).contiguous().to(dev)
        dw = torch.empty(cout, k, k, cin, device=dev)
        ops.conv2d_wgrad(img(xd), img(gyd), k, s, dw)
        check(f"wgrad {cin}->{cout} k{k} s{s} rep{rep}", dw.permute(0, 3, 1, 2), wt.grad)
# pair-view input (encoder first layers) and channel-slice dy (u3 of UpsampleSkip)
B, h, w = 2, 16, 16
cat = torch.randn(B, h, w, 64, generator=g)
wt = (torch.randn(64, 32, 3, 3, generator=g) / 17).requires_grad_(True)
xp = torch.cat([cat[..., :32], cat[..., 32:]], 0).permute(0, 3, 1, 2).contiguous()
y = F.conv2d(xp, wt, None, padding=1); gy = torch.randn(y.shape, generator=g); y.backward(gy)
poison()
catd = cat.to(dev); gyd = gy.permute(0, 2, 3, 1).contiguous().to(dev)
dw = torch.empty(64, 3, 3, 32, device=dev)
ops.conv2d_wgrad(img_pair(catd, 32), img(gyd), 3, 1, dw)
check("wgrad pair-view x", dw.permute(0, 3, 1, 2), wt.grad)
# GRU-style row GEMM with row masks, T images, x with img_stride 0
T, BN, N = 3, 4 * 100, 100
counts = torch.tensor([100, 37, 0, 64], dtype=torch.int32)
valid = (torch.arange(N)[None, :] < counts[:, None]).reshape(-1)
dy = torch.randn(T, BN, 128, generator=g); xr = torch.randn(BN, 64, generator=g); hpl = torch.randn(T, BN, 128, generator=g)
want_h = sum((dy[t][valid].T @ hpl[t][valid]) for t in range(T)); want_x = sum((dy[t][valid].T @ xr[valid]) for t in range(T))
poison()
nan = float("nan")
dyd = dy.clone(); dyd[:, ~valid] = nan; hd = hpl.clone(); hd[:, ~valid] = nan; xd = xr.clone(); xd[~valid] = nan
dyd, hd, xd, cd = dyd.to(dev), hd.to(dev), xd.to(dev), counts.to(dev)
mk = lambda t, n_img, c, stride: DfImg(t.data_ptr(), n_img, 1, BN, c, c, n_img, stride, 0)
dW = torch.empty(128, 192, device=dev)
ops.conv2d_wgrad(mk(hd, T, 128, BN * 128), mk(dyd, T, 128, BN * 128), 1, 1, dW, ld_co=192, dw_off=0, row_counts=cd, rows_per_seg=N)
ops.conv2d_wgrad(mk(xd, T, 64, 0), mk(dyd, T, 128, BN * 128), 1, 1, dW, ld_co=192, dw_off=128, row_counts=cd, rows_per_seg=N)
check("row-gemm h part", dW[:, :128], want_h); check("row-gemm x part", dW[:, 128:], want_x)
# generic 1x1 kernel (64 output channels) with row masks and a 32-channel x (the decoder's dW1^T GEMM)
dp = torch.randn(BN, 32, generator=g); want = (xr[valid].T @ dp[valid])
poison()
dpd = dp.clone(); dpd[~valid] = nan; dpd = dpd.to(dev)
dWt = torch.empty(64, 32, device=dev)
ops.conv2d_wgrad(mk(dpd, 1, 32, BN * 32), mk(xd, 1, 64, BN * 64), 1, 1, dWt, ld_co=32, dw_off=0, row_counts=cd, rows_per_seg=N)
check("row-gemm generic 1x1, K=32, masked", dWt, want)
dWt2 = torch.empty(128, 32, device=dev)
want2 = hpl[0][valid].T @ dp[valid]
ops.conv2d_wgrad(mk(dpd, 1, 32, BN * 32), mk(hd, 1, 128, BN * 128), 1, 1, dWt2, ld_co=32, dw_off=0, row_counts=cd, rows_per_seg=N)
check("row-gemm 128-tile 1x1, K=32, masked", dWt2, want2)
