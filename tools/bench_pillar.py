"""Pillarise stage alone (eval, both clouds as one 2B-sample set): us per pair and per-kernel split via HIP events.
usage: python tools/bench_pillar.py [B]   (DF_P2_DBG=1|2|4 ablate the band kernel: no pillar loop / no zero stores / no sort)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deflow_amd
from deflow_amd._lib import DfImg
from deflow_amd.synth import synth_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda")
m = deflow_amd.DeFlow().to(dev).eval()
b = synth_batch(B, 80000, device=dev)
pts = torch.cat([b["pc0"], b["pc1"]], 0).contiguous()
emb = m.embedder
bstar = torch.empty(B, 512, 512, 64, device=dev)
d = DfImg(bstar.data_ptr(), 2 * B, 512, 512, 32, 64, B, bstar.stride(0), 32)
for train in (False, True):
    emb.train(train)
    with torch.no_grad():
        for _ in range(3):
            emb.pillarize(pts, d, train, need_cells=False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            emb.pillarize(pts, d, train, need_cells=False)
        e1.record()
        torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3 / B
    print(f"B={B} train={train} DF_P2_DBG={os.environ.get('DF_P2_DBG', '0')}: {us:.2f} us/pair  ({69.03e6 / (us * 1e-6) / 8e12:.3f} of 8 TB/s)")
