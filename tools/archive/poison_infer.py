"""uninitialised-memory / determinism probe: forward results must not depend on what the allocator's recycled blocks contain"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, deflow_amd
from deflow_amd.synth import synth_batch
dev = torch.device("cuda"); torch.manual_seed(0)
for train in (False, True):
    m = deflow_amd.DeFlow().to(dev).train(train)
    batch = synth_batch(2, 80000, device=dev)
    with torch.no_grad():
        a = m.forward_padded(batch)["flow"].clone()
        a2 = m.forward_padded(batch)["flow"].clone()
        m.last_state = None
        torch.cuda.empty_cache()
        junk = torch.full((6 * 1024 ** 3,), float("nan"), device=dev)   # 24 GB of NaN, then back to the allocator
        del junk
        b = m.forward_padded(batch)["flow"].clone()
        cnt = m.last_state["counts0"]
    for i in range(2):
        n = int(cnt[i])
        print("train" if train else "eval", "sample", i, "valid rows", n, "repeat max|diff|", float((a[i, :n] - a2[i, :n]).abs().max()),
              "poisoned max|diff|", float((a[i, :n] - b[i, :n]).abs().max()), "nan rows", int(torch.isnan(b[i, :n]).any(1).sum()))
