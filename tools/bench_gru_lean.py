"""micro-benchmark of the lean GRU decoder kernels (forward with planes, backward data pass, weight-gradient pass) at the bench
shape (B=16, N=80000 padded rows, ~90 % valid, 512x512), straight through the C ABI.  DF_LIB=<variant .so> selects a build;
GRU_GEN=6 calls the ring generation (df_gru_ring_*) when the library has it."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deflow_amd import ops
from deflow_amd.decoder import ConvGRUDecoder, PointSet
from deflow_amd._lib import DfGruWeightsT, call, img, ptr, stream

dev = torch.device("cuda")
B, N, H, T = int(os.environ.get("GRU_B", 16)), 80000, 512, 4
GEN = int(os.environ.get("GRU_GEN", "4"))
torch.manual_seed(0)
head = ConvGRUDecoder(num_iters=T).to(dev)
before = torch.randn(B, H, H, 64, device=dev)
after = torch.randn(B, H, H, 64, device=dev)
coords = torch.zeros(B, N, 3, dtype=torch.int32, device=dev)
coords[..., 1:] = torch.randint(0, H, (B, N, 2), device=dev, dtype=torch.int32)
offs = (torch.rand(B, N, 3, device=dev) - 0.5) * 0.2
counts = torch.full((B,), int(N * 0.98), dtype=torch.int32, device=dev)
ps = PointSet(coords, offs, counts)
BN = B * N
f32 = dict(dtype=torch.float32, device=dev)


def timeit(fn, n=6):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


W0, keep = head._weights()
w_zr, b_zr, w_q = keep
xtab = head._xtab(W0)
W, keep2 = head._weights_x2(W0, keep)
w1 = head.decoder[0].weight.detach()
wt_zr = head._split_x2(ops.weight_transpose(w_zr.view(256, 1, 1, 192)).view(192, 256))
wt_q = head._split_x2(ops.weight_transpose(w_q.view(128, 1, 1, 192)).view(192, 128))
wt_1 = head._split_x2(ops.weight_transpose(w1.view(32, 1, 1, 192)).view(192, 32))
WT = DfGruWeightsT(ptr(wt_zr), ptr(wt_q), ptr(wt_1))
flow = torch.empty(B, N, 3, **f32)
hs = torch.empty((T + 1) * BN * 128, **f32)
gpl = torch.empty(4 * T * BN * 128, **f32)
dh0, dpre1 = torch.empty(BN, 128, **f32), torch.empty(BN, 32, **f32)
dflow = torch.randn(B, N, 3, **f32)
PW = call("df_gru_lean_partial_width")
FWD = "df_gru_ring_fwd" if GEN == 6 else "df_gru_lean_fwd"
BWD = "df_gru_ring_bwd" if GEN == 6 else "df_gru_lean_bwd"
tile = 128 if GEN == 6 else 64
partial = torch.zeros(B * ((N + tile - 1) // tile), PW, **f32)


def fwd(save=True):
    call(FWD, img(before), img(after), ptr(coords), ptr(offs), ptr(counts), B, N, T, W, ptr(xtab), ptr(flow), ptr(hs) if save else None, 3, stream())


def bwd():
    call(BWD, ptr(dflow), ptr(offs), ptr(counts), B, N, T, W, WT, ptr(xtab), ptr(hs), ptr(gpl), ptr(dh0), ptr(dpre1), ptr(partial), 3,
         stream())


nsplit = call("df_gru_wgrad_splits")
ws = torch.empty(nsplit, 384, 128, **f32)


def wgrad():
    call("df_gru_lean_wgrad", ptr(hs), ptr(gpl), ptr(counts), B, N, T, ptr(ws), nsplit, 3, stream())


t_f = timeit(fwd)
t_i = timeit(lambda: fwd(False))
fl = flow.clone()
t_b = timeit(bwd)
t_w = timeit(wgrad)
torch.cuda.synchronize()
chk = dict(flow=float(fl[:, : int(N * 0.98)].double().abs().sum()), dh0=float(dh0.view(B, N, 128)[:, : int(N * 0.98)].double().abs().sum()),
           ws=float(ws.double().abs().sum()))
print(f"lib={os.environ.get('DF_LIB', 'default')} gen={GEN} B={B}: fwd(save) {t_f:.3f} ms  fwd(no save) {t_i:.3f} ms  bwd {t_b:.3f} ms  "
      f"wgrad {t_w:.3f} ms  trio {t_f + t_b + t_w:.3f} ms  checks {chk}", flush=True)
