"""robustness sweep: the sparse edge kernels vs the dense kernels on odd batch sizes / point counts / an empty sample"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, deflow_amd
from deflow_amd.optim import Trainer
from deflow_amd.synth import synth_batch
dev = torch.device("cuda")
worst = 0.0
for (B, N, grid) in [(1, 1000, 64), (3, 4097, 128), (5, 777, 64), (2, 20000, 256), (1, 80000, 512)]:
    torch.manual_seed(B * 7 + N)
    rng = [-0.1 * grid, -0.1 * grid, -3, 0.1 * grid, 0.1 * grid, 3]
    m = deflow_amd.DeFlow(grid_feature_size=[grid, grid], point_cloud_range=rng).to(dev).train()
    tr = Trainer(m, lr=2e-4)
    batch = synth_batch(B, N, grid_hw=(grid, grid), device=dev)
    if B >= 3:   # one sample without any valid point, one with very few
        batch["pc0"][1] = float("nan")
        batch["pc0"][2, 5:] = float("nan")
    out = {}
    for mode in ("0", "1"):
        os.environ["DF_DENSE_CANVAS_GRAD"] = mode
        tr.flat.zero_grad(); tr.sink.begin()
        m.forward_padded(batch)
        loss = tr.loss_on_last_forward(batch)
        loss.backward()
        out[mode] = (tr.flat.grad.clone(), float(loss.detach()))
    os.environ.pop("DF_DENSE_CANVAS_GRAD")
    g0, g1 = out["0"][0], out["1"][0]
    err = float((g0 - g1).abs().max() / (g1.abs().max() + 1e-30))
    worst = max(worst, err)
    print(f"B={B} N={N} grid={grid}: loss sparse {out['0'][1]:.6f} dense {out['1'][1]:.6f}  max|dgrad|/max|grad| = {err:.2e}  finite={bool(torch.isfinite(g0).all())}")
print("worst", worst)
assert worst < 1e-4
