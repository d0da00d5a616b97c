"""bf16 inference path (UNet on bf16 MFMA) vs the fp32 path: accuracy and speed; BASELINE configs[1] and configs[4] shapes"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, deflow_amd
from deflow_amd.synth import synth_batch
dev = torch.device("cuda")
out = {}
for tag, kw, B, N, grid in [("cfg1_512_80k_4it", dict(), 1, 80000, 512), ("cfg1_B16", dict(), 16, 80000, 512),
                            ("cfg4_1024_160k_8it", dict(grid_feature_size=[1024, 1024], point_cloud_range=[-102.4, -102.4, -3, 102.4, 102.4, 3], num_iters=8), 1, 160000, 1024)]:
    torch.manual_seed(0)
    m = deflow_amd.DeFlow(**kw).to(dev).eval()
    batch = synth_batch(B, N, grid_hw=(grid, grid), device=dev)
    res = {}
    with torch.no_grad():
        for dt in ("fp32", "bf16"):
            m.inference_dtype = dt
            for _ in range(3): st = m.forward_padded(batch)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = 10
            for _ in range(n): st = m.forward_padded(batch)
            torch.cuda.synchronize()
            res[dt] = ((time.perf_counter() - t0) / n * 1e3, st["flow"].clone(), int(st["counts0"][0]))
    nv = res["fp32"][2]
    f32, f16 = res["fp32"][1][0, :nv], res["bf16"][1][0, :nv]
    err = float((f32 - f16).abs().max()); ref = float(f32.abs().max())
    rms = float((f32 - f16).pow(2).mean().sqrt())
    out[tag] = {"fp32_ms": res["fp32"][0], "bf16_ms": res["bf16"][0], "speedup": res["fp32"][0] / res["bf16"][0],
                "max_abs_flow_err": err, "rms_flow_err": rms, "max_abs_flow": ref}
    print(tag, json.dumps(out[tag]))
