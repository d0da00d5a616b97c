"""Is the training step's gradient arena bit-reproducible on the SMALL test configuration (64 x 64 grid, 2 x 1500 points)?
Repeats forward + loss + backward R times in each mode and lists the parameters whose gradient ever differs from the first
repetition, plus (DF_PROBE_DEEP=1) a hash of the intermediate tensors feeding the pillar feature net's backward.

    python tools/grad_repro_probe.py [reps] [fp32|bf16] [graph]"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import torch
from deflow_amd.optim import Trainer
from test_gpu_model import build_pair, make_batch, to_dev

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
graph = len(sys.argv) > 3 and sys.argv[3] == "graph"
dev = torch.device("cuda", 0)
if os.environ.get("DF_PROBE_FULL"):       # the bench's own shape: B pairs x 80 000 points on the 512 x 512 grid
    import deflow_amd
    from deflow_amd.synth import synth_batch
    torch.manual_seed(0)
    model = deflow_amd.DeFlow().to(dev)
    batch = synth_batch(int(os.environ["DF_PROBE_FULL"]), 80000, device=dev)
else:
    _, model = build_pair(dev, 41, decoder_option="gru", num_iters=2)
    batch = to_dev(make_batch(2, 1500, 7000), dev)
model.train()
tr = Trainer(model, lr=0.0, dtype=dtype)      # lr 0: parameters never move, every repetition is the same problem
if graph:
    tr.capture(batch)
    step = lambda: tr.step_captured()
else:
    step = lambda: tr.step(batch)
from deflow_amd import autograd as AG
taps, tap_first, tap_bad = {}, {}, {}


def tap(stage, **kw):
    for k, v in kw.items():
        if k.startswith(("pts_sorted", "key_sorted")):     # rows past the valid total are never written
            n = int(kw["counts" + k[-1]].sum())
            v = v[:n]
        taps[f"{stage}.{k}"] = v.clone()


if os.environ.get("DF_PROBE_DEEP") == "1" and not graph:
    AG.TAP = tap
# DF_STRESS_THREAD=<kind> (round 4): a neighbour THREAD of this process on its own stream (tools/pfn_neighbour.py kinds, DF_NB_ONLY
# filter) -- is the whole training step bit-reproducible beside it?
if os.environ.get("DF_STRESS_THREAD"):
    import threading, time as _t
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import pfn_neighbour
    _stop, _count = threading.Event(), [0]

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
    threading.Thread(target=_nb, daemon=True).start()
    _t.sleep(8.0)
first, bad = None, {}
for r in range(reps):
    step()
    torch.cuda.synchronize()
    g = tr.flat.grad.clone()
    for k, v in taps.items():
        if k not in tap_first:
            tap_first[k] = v
        elif not torch.equal(v, tap_first[k]):
            d = (v - tap_first[k]).abs()
            where = ""
            if k.endswith("dh0"):
                continue
            if False:
                where = f" cols<64: {float(d[:, :64].max()):.2e} cols>=64: {float(d[:, 64:].max()):.2e}"
            elif v.dim() == 4 and v.shape[3] == 64:
                where = f" ch<32: {float(d[..., :32].max()):.2e} ch>=32: {float(d[..., 32:].max()):.2e}"
            tap_bad.setdefault(k, []).append(f"rep {r} n={int((d > 0).sum())} max {float(d.max()):.2e}{where}")
    if first is None:
        first = g
        continue
    if not torch.equal(g, first):
        for n, p in tr.flat.named:
            off, k = tr.flat.slots[n]
            if not torch.equal(g[off:off + k], first[off:off + k]):
                d = float((g[off:off + k] - first[off:off + k]).abs().max() / first[off:off + k].abs().max().clamp_min(1e-30))
                bad.setdefault(n, []).append((r, d))
if os.environ.get("DF_STRESS_THREAD"):
    _stop.set()
    _t.sleep(0.5)
print(f"[lib {os.path.basename(os.environ.get('DF_LIB') or 'libdeflow_amd.so')}, thread {os.environ.get('DF_STRESS_THREAD', 'none')} /{os.environ.get('DF_NB_ONLY', '.')}/] "
      f"{dtype} graph={graph} reps={reps}: {len(bad)} parameters with a varying gradient")
for n, v in bad.items():
    print(f"   {n}: {len(v)} of {reps - 1} repetitions differ, max rel diff {max(d for _, d in v):.2e}")
for k, v in tap_bad.items():
    print(f"   tap {k}: {len(v)} repetitions differ; e.g. {v[:3]}")
