#!/bin/bash
# A/B of environment switches on the training bench: tools/env_ab.sh "NAME=VAL ..." "NAME=VAL ..." ...  (each argument = one leg's
# environment, "" = defaults); legs run in the given order, ENV_AB_REPS times round-robin; prints step and GRU / pillar stage times
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for rep in $(seq ${ENV_AB_REPS:-2}); do
  i=0
  for leg in "$@"; do
    i=$((i+1))
    env $leg python bench.py --steps ${ENV_AB_STEPS:-8} --warmup 3 --no-extras --no-cpu-baseline --no-loader > gpurun_out/env_ab_$i.json 2> gpurun_out/env_ab_$i.err
    python - "$i" "$leg" <<'PY'
import json, sys
i, leg = sys.argv[1], sys.argv[2]
try:
    d = json.loads([l for l in open(f"gpurun_out/env_ab_{i}.json") if l.startswith("{")][-1])
    h = d.get("roofline_hbm", {})
    print(f"[{leg or 'default':28s}] step {d['ms_per_step']:.2f} ms  gru fwd/bwd/wgrad " + " / ".join(f"{h[k]['ms_per_step']:.3f}" for k in ("gru_fwd", "gru_bwd", "gru_wgrad") if k in h)
          + (f"  pillarise_fwd {h['pillarise_fwd']['ms_per_step']:.3f} ms frac {h['pillarise_fwd'].get('frac_hbm', h['pillarise_fwd'].get('frac', 0)):.3f}" if "pillarise_fwd" in h else "")
          + f"  canvas {d.get('pillar_canvas')}", flush=True)
except Exception as e:
    print(leg, "FAILED", e, open(f"gpurun_out/env_ab_{i}.err").read()[-1500:])
PY
  done
done
