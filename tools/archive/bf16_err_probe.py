"""Where does the bf16 training mode's gradient error come from?  (7.7e-2 rms on one tensor.)
B = 16 @ 256 x 256 step: per parameter gradient, rms-relative error of the bf16 step against the fp32 HIP step (itself within 1e-5 of
float64), for the mode variants selected by environment switches.  usage: python tools/bf16_err_probe.py [tag]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, deflow_amd
from deflow_amd import ops
from deflow_amd.optim import Trainer
from deflow_amd.synth import synth_batch
dev = torch.device("cuda")
grid = int(os.environ.get("PG", "256")); npts = 20000 if grid == 256 else 80000
half = 0.1 * grid
cfg = dict(voxel_size=[0.2, 0.2, 6], point_cloud_range=[-half, -half, -3, half, half, 3], grid_feature_size=[grid, grid])
batch = synth_batch(16, npts, seed=4242, grid_hw=(grid, grid), device=dev)
def run(dtype):
    torch.manual_seed(16)
    m = deflow_amd.DeFlow(**cfg).to(dev).train()
    tr = Trainer(m, lr=0.0, dtype=dtype)
    tr.flat.zero_grad(); tr.sink.begin()
    with ops.mfma_bf16(tr.mfma_bf16, tr.bf16_store):
        loss = tr._forward_backward(batch)
    torch.cuda.synchronize()
    return float(loss), {k: p.grad.detach().double().clone() for k, p in m.named_parameters()}
l32, g32 = run("fp32")
l16, g16 = run("bf16")
rows = []
for k in g32:
    if k.endswith(".conv.bias") and "encoder" in k:
        continue
    n = float(g32[k].norm())
    rows.append((float((g16[k] - g32[k]).norm()) / max(n, 1e-300), k))
rows.sort(reverse=True)
print(f"[{sys.argv[1] if len(sys.argv) > 1 else 'default'}] loss fp32 {l32:.6f} bf16 {l16:.6f}; rms-relative error of the bf16 step's gradients vs the fp32 step, worst first")
for e, k in rows[:14]:
    print(f"   {e:.3e}  {k}")
print("   median", rows[len(rows) // 2][0])
if os.environ.get("PROBE_AUTOCAST") == "1":
    # the same step of the ORACLE under torch.autocast(bfloat16) on the CPU (what Lightning's precision="bf16-mixed" does to the reference):
    # its gradient error against the fp32 oracle, per tensor, beside the HIP bf16 mode's
    from oracle import ref_torch as O
    torch.manual_seed(16)
    ref = O.DeFlow(**cfg).train()
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    bc = {k: v.cpu() for k, v in batch.items()}
    def orun(ac):
        m = O.DeFlow(**cfg); m.load_state_dict(sd); m.train()
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=ac):
            res = m(bc); loss = O.training_loss(res, bc)
        loss.backward()
        return float(loss.detach()), {k: p.grad.double() for k, p in m.named_parameters()}
    lo, go = orun(False)
    la, ga = orun(True)
    r2 = []
    for k in go:
        if k.endswith(".conv.bias") and "encoder" in k:
            continue
        n = float(go[k].norm())
        r2.append((float((ga[k] - go[k]).norm()) / max(n, 1e-300), float((g16[k].cpu() - go[k]).norm()) / max(n, 1e-300), k))
    r2.sort(reverse=True)
    print(f"oracle loss fp32 {lo:.6f} autocast-bf16 {la:.6f}; per tensor: [oracle under autocast(bf16) vs oracle fp32] | [HIP bf16 mode vs oracle fp32]")
    for a, h, k in r2[:14]:
        print(f"   {a:.3e} | {h:.3e}  {k}")
    print("   median", r2[len(r2) // 2][0], "| HIP", sorted(x[1] for x in r2)[len(r2) // 2])
