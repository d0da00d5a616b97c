"""one-line digests of bench.py JSON lines:  python tools/show_bench.py a.json b.json ..."""
import json
import sys
for path in sys.argv[1:]:
    d = json.loads([l for l in open(path) if l.startswith("{")][-1])
    g = lambda *ks: (lambda v: round(v, 3) if isinstance(v, float) else v)(__import__("functools").reduce(lambda a, k: a.get(k, {}) if isinstance(a, dict) else {}, ks, d))
    print(path, "| fp32 ms", g("ms_per_step"), "pairs/s", g("value"), "| bf16 ms", g("bf16_training", "ms_per_step"), "| graph ms", g("hip_graph", "ms_per_step"),
          "host", g("hip_graph", "host_ms_per_step"), "| cfg4 bf16", g("bf16_training", "configs4_shape", "bf16_ms_per_step"), "| fwd B1", g("forward_only", "ms_per_pair"),
          "bf16", g("forward_only", "bf16_ms_per_pair"), "| cfg4 inf", g("bf16_inference", "ms_per_pair"))
    print("    dom", g("roofline", "kernel"), g("roofline", "achieved"), "| hbm:", {k: (round(v["ms_per_step"], 2) if "ms_per_step" in v else round(v["us_per_pair"], 1)) for k, v in d.get("roofline_hbm", {}).items()})
