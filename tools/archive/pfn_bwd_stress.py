"""Isolated stress of the pillar feature net's backward (df_pfn_bwd_stats / finalize / weights + colsum) on FIXED inputs:
is the result bit-reproducible while another process keeps the GPU busy?   python tools/pfn_bwd_stress.py [reps]"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import torch
from deflow_amd._lib import img
from test_gpu_model import build_pair, make_batch, to_dev

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = torch.device("cuda", 0)
_, model = build_pair(dev, 41, decoder_option="gru", num_iters=2)
model.train()
batch = to_dev(make_batch(2, 1500, 7000), dev)
with torch.no_grad():
    st = model.forward_padded(batch, engine_tape=True)["engine"]
emb = model.embedder
g = torch.Generator(device="cpu").manual_seed(3)
dbstar = torch.randn(2, 64, 64, 64, generator=g).to(dev)
from deflow_amd._lib import call, ptr, stream
first, nbad, worst = None, 0, 0.0
stage_first, stage_bad = {}, {}


def bwd_one(pst, gout, acc, dW, dgamma, dbeta, tag):
    """= DynamicEmbedder.pillarize_bwd, with every intermediate kept for the bisection"""
    B, N, _ = pst.pts.shape
    gm, s = emb.geom, stream()
    w = emb._lin.weight.detach()
    nbs = max(1, min(256, (N + 31) // 32))
    partial = torch.empty(B, nbs, 32, 2, device=dev)
    call("df_pfn_bwd_stats", ptr(pst.pts_sorted), ptr(pst.cell_rng), ptr(pst.key_sorted), ptr(pst.counts), B, gm, ptr(w), ptr(pst.bn_ss),
         pst.bn_stride, emb.mode, gout, ptr(partial), nbs, s)
    coef = torch.empty(B, 2, 32, device=dev)
    call("df_pfn_bwd_finalize", ptr(partial), B, nbs, ptr(pst.counts), ptr(dgamma), ptr(dbeta), int(acc), ptr(coef), s)
    dwp = torch.empty(B * nbs, 288, device=dev)
    call("df_pfn_bwd_weights", ptr(pst.pts_sorted), ptr(pst.cell_rng), ptr(pst.key_sorted), ptr(pst.counts), B, gm, ptr(w), ptr(pst.bn_ss),
         pst.bn_stride, emb.mode, ptr(coef), gout, ptr(dwp), nbs, s)
    call("df_colsum_finalize", ptr(dwp), B * nbs, 288, 1, ptr(dW), int(acc), s)
    return {f"{tag}.partial": partial, f"{tag}.coef": coef, f"{tag}.dwp": dwp, f"{tag}.dgamma": dgamma.clone(), f"{tag}.dW": dW.clone()}


# DF_STRESS_SIDE=matmul: the bf16-MFMA neighbour INSIDE this process -- a side stream kept busy with bf16 GEMMs (round 4: does the
# failure need a second PROCESS, i.e. the GPU's time-slicing between contexts, or just a concurrent bf16-MFMA kernel?)
side = torch.cuda.Stream() if os.environ.get("DF_STRESS_SIDE") else None
if side is not None:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    kind = os.environ["DF_STRESS_SIDE"]
    if kind.startswith("poison"):           # poison[:what] -- tools/poison.hip: NaN in 240 VGPRs and all of LDS on every CU
        import ctypes
        _pl = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "libpoison.so"))
        _what = int(kind.split(":")[1]) if ":" in kind else 7
        side_step = lambda: _pl.poison_launch(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), _what)
    elif kind.startswith("mfma"):           # mfma:<k> -- tools/pk_mfma_hazard.hip's register-resident MFMA stream (k: 0 = 16x16x32_bf16,
        import ctypes                       # 1 = 32x32x16_bf16, 2 = 16x16x4_f32, 3 = 16x16x32_f16), ~2 ms per launch on half the chip
        _hz = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "libhazard.so"))
        _k = int(kind.split(":")[1])
        side_step = lambda: _hz.hazard_neighbour_launch(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), _k, 20000, 512)
    else:
        import pfn_neighbour
        side_step = pfn_neighbour.make({"matmul": "matmul_bf16"}.get(kind, kind), dev)
    torch.cuda.synchronize()
# DF_STRESS_THREAD=<kind>: the neighbour as a second host THREAD of this process with its own stream (tools/pfn_neighbour.py kinds).
# A side stream fed by THIS thread does not overlap with the victim at the small test size: the GPU drains each tiny kernel before
# the host has enqueued the next, so the two streams alternate.  Two host threads enqueue at the same time, like two processes do.
if os.environ.get("DF_STRESS_THREAD"):
    import threading
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import pfn_neighbour
    _stop = threading.Event()
    _count = [0]

    def _nb():
        torch.cuda.set_device(dev)
        st_ = torch.cuda.Stream()
        with torch.cuda.stream(st_):
            step_ = pfn_neighbour.make(os.environ["DF_STRESS_THREAD"], dev)
            while not _stop.is_set():
                step_()
                _count[0] += 1
                if _count[0] % 8 == 0:
                    st_.synchronize()
    _th = threading.Thread(target=_nb, daemon=True)
    _th.start()
    import time as _t
    _t.sleep(8.0)
for r in range(reps):
    if side is not None:
        with torch.cuda.stream(side):
            side_step()                     # (each kind enqueues >= the ~0.3 ms of one repetition: the side stream stays busy)
    dW, dgamma, dbeta = torch.empty(32, 9, device=dev), torch.empty(32, device=dev), torch.empty(32, device=dev)
    inter = bwd_one(st["p0"], img(dbstar, 32, 0), False, dW, dgamma, dbeta, "c0")
    inter.update(bwd_one(st["p1"], img(dbstar, 32, 32), True, dW, dgamma, dbeta, "c1"))
    out = (dW, dgamma, dbeta)
    torch.cuda.synchronize()
    for k, v in inter.items():
        if k not in stage_first:
            stage_first[k] = v.clone()
        elif not torch.equal(v, stage_first[k]):
            d = (v - stage_first[k]).abs()
            stage_bad.setdefault(k, []).append(f"rep {r}: n={int((d > 0).sum())} max {float(d.max()):.2e} idx {d.reshape(-1).argmax().item()}")
    cur = torch.cat([o.reshape(-1) for o in out]).clone()
    if first is None:
        first = cur
    elif not torch.equal(cur, first):
        nbad += 1
        worst = max(worst, float((cur - first).abs().max() / first.abs().max()))
print(f"[lib {os.path.basename(os.environ.get('DF_LIB', 'libdeflow_amd.so'))}, side stream {os.environ.get('DF_STRESS_SIDE', 'none')}, thread {os.environ.get('DF_STRESS_THREAD', 'none')}] pfn backward: {nbad} of {reps - 1} repetitions differ from the first (worst rel {worst:.2e})")
if os.environ.get("DF_STRESS_THREAD"):
    _stop.set(); _th.join(10.0)
    print(f"   neighbour thread ran {_count[0]} steps")
for k, v in stage_bad.items():
    print(f"   {k}: {len(v)} reps differ, e.g. {v[:4]}")
