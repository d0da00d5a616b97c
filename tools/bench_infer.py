"""BASELINE configs[1]: deflow forward-only inference, synthetic 80k-pt pair, 512x512x64 BEV, 4 GRU iterations."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deflow_amd
from deflow_amd.synth import synth_batch
dev = torch.device("cuda")
torch.manual_seed(0)
m = deflow_amd.DeFlow().to(dev).eval()
m.inference_dtype = os.environ.get("DF_INFER_DTYPE", "fp32")      # "bf16": UNet + GRU GEMMs on bf16 MFMA
out = {"dtype": m.inference_dtype}
for B in (1, 4, 16):
    batch = synth_batch(B, 80000, device=dev)
    with torch.no_grad():
        for _ in range(3):
            m.forward_padded(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20 if B == 1 else 8
        for _ in range(n):
            m.forward_padded(batch)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
    out[f"B{B}"] = {"ms": dt * 1e3, "pairs_per_s": B / dt, "tflops": 391.6e9 * B / dt / 1e12}
# B=1 again as a captured HIP graph: the forward is sync-free and allocation-stable, so the whole launch sequence
# (~85 kernels) replays with one host call
try:
    batch = synth_batch(1, 80000, device=dev)
    with torch.no_grad():
        s_ = torch.cuda.Stream()
        s_.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s_):
            for _ in range(3):
                m.forward_padded(batch)
        torch.cuda.current_stream().wait_stream(s_)
        ref = m.forward_padded(batch)["flow"].clone()   # eager result
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            m.forward_padded(batch)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            g.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 50
        nv = int(m.last_state["counts0"][0])   # rows past the valid count are never written (padding)
        same = bool(torch.equal(m.last_state["flow"][0, :nv], ref[0, :nv]))
    out["B1_graph"] = {"ms": dt * 1e3, "pairs_per_s": 1 / dt, "tflops": 391.6e9 / dt / 1e12, "replay_equals_eager": same}
except Exception as e:  # noqa
    out["B1_graph"] = {"error": repr(e)[:300]}
print(json.dumps(out))
