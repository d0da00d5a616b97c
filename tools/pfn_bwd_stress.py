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


for r in range(reps):
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
print(f"pfn backward alone: {nbad} of {reps - 1} repetitions differ from the first (worst rel {worst:.2e})")
for k, v in stage_bad.items():
    print(f"   {k}: {len(v)} reps differ, e.g. {v[:4]}")
