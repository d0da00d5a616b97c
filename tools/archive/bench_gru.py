"""micro-benchmark of the fused decoder kernels at the bench shape (B=16, N=80000, 512x512)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deflow_amd.decoder import ConvGRUDecoder, PointSet
from deflow_amd._lib import img

dev = torch.device("cuda")
B, N, H = 16, 80000, 512
torch.manual_seed(0)
head = ConvGRUDecoder(num_iters=4).to(dev)
before = torch.randn(B, H, H, 64, device=dev)
after = torch.randn(B, H, H, 64, device=dev)
coords = torch.zeros(B, N, 3, dtype=torch.int32, device=dev)
coords[..., 1:] = torch.randint(0, H, (B, N, 2), device=dev, dtype=torch.int32)
offs = (torch.rand(B, N, 3, device=dev) - 0.5) * 0.2
counts = torch.full((B,), int(N * 0.9), dtype=torch.int32, device=dev)
ps = PointSet(coords, offs, counts)


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


flops = 0.9 * B * N * (4 * 3 * 2 * 192 * 128 + 2 * 192 * 32)
t_inf = timeit(lambda: head.run(img(before), img(after), ps, False))
t_trn = timeit(lambda: head.run(img(before), img(after), ps, True))
print(f"gru fwd inference {t_inf:.2f} ms ({flops / t_inf / 1e9:.1f} TF/s)   training(save) {t_trn:.2f} ms ({flops / t_trn / 1e9:.1f} TF/s)")

# backward data pass alone (needs a forward save each time because it overwrites planes; time fwd+bwd and subtract)
def fwd_bwd():
    flow, sv = head.run(img(before), img(after), ps, True)
    f32 = dict(dtype=torch.float32, device=dev)
    BN = B * N
    from deflow_amd import ops
    from deflow_amd._lib import DfGruWeightsT, call, ptr, stream
    W, keep = head._weights()
    w_zr, b_zr, w_q = keep
    w1 = head.decoder[0].weight.detach()
    wt_zr = ops.weight_transpose(w_zr.view(256, 1, 1, 192)).view(192, 256)
    wt_q = ops.weight_transpose(w_q.view(128, 1, 1, 192)).view(192, 128)
    wt_1 = ops.weight_transpose(w1.view(32, 1, 1, 192)).view(192, 32)
    WT = DfGruWeightsT(ptr(wt_zr), ptr(wt_q), ptr(wt_1))
    dh0, dx = torch.empty(BN, 128, **f32), torch.empty(BN, 64, **f32)
    dpre1, xbuf = torch.empty(BN, 32, **f32), torch.empty(BN, 64, **f32)
    bp = torch.zeros(B * ((N + 63) // 64), 772, **f32)
    call("df_gru_decoder_bwd", ptr(flow), ptr(ps.offs), ptr(ps.counts), B, N, 4, W, WT, ptr(sv), ptr(dh0), ptr(dx),
         ptr(dpre1), ptr(xbuf), ptr(bp), stream())
t_fb = timeit(fwd_bwd, 4)
print(f"gru fwd(train)+bwd data pass {t_fb:.2f} ms -> bwd ~ {t_fb - t_trn:.2f} ms (DF_GRU_DBG={os.environ.get('DF_GRU_DBG', '0')})")

# weight-gradient GEMMs of the three gates (after a backward pass left the gate gradients in the planes)
def wgrad_fused():
    from deflow_amd._lib import call, ptr, stream
    nsplit = call("df_gru_wgrad_splits")
    ws = torch.empty(nsplit, 384, 192, device=dev)
    call("df_gru_wgrad", ptr(SV), ptr(XB), ptr(ps.counts), B, N, 4, ptr(ws), nsplit, stream())
    out = torch.empty(384, 192, device=dev)
    call("df_conv2d_wgrad_reduce", ptr(ws), nsplit, 384, 1, 192, ptr(out), 192, 0, stream())
    return out
_, SV = head.run(img(before), img(after), ps, True)
XB = torch.randn(B * N, 64, device=dev)
t_wg = timeit(wgrad_fused)
wf = 0.9 * B * N * 4 * 2 * 192 * 384
print(f"gru fused weight gradients {t_wg:.2f} ms ({wf / t_wg / 1e9:.1f} TF/s)")
