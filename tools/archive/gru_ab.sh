#!/bin/bash
# A/B of the GRU forward's plane stores (round 4, second session): same bench, three libraries
#   gru_base   -DDF_GRU_MERGE_SAVES=0 -DDF_GRU_PERM_PLANES=0   (the round-4 record's kernels)
#   gru_merge  -DDF_GRU_PERM_PLANES=0                           (two store groups per iteration instead of four)
#   default    both: + z, r, q as register-order tiles (16-byte stores / loads)
# apply tools/gru_store_ab.patch, then build the variants (CPU): python -c "from deflow_amd import build as B; B.build_variant('gru_base', [...]); ..."
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for v in ${GRU_AB_VARIANTS:-gru_base gru_merge default gru_base default}; do
  if [ "$v" = default ]; then unset DF_LIB; else export DF_LIB=$PWD/deflow_amd/_build/$v/lib$v.so; fi
  python bench.py --steps ${GRU_AB_STEPS:-8} --warmup 3 --no-extras --no-cpu-baseline --no-loader > gpurun_out/gru_ab_$v.json 2> gpurun_out/gru_ab_$v.err
  python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
try:
    d = json.loads([l for l in open(f"gpurun_out/gru_ab_{v}.json") if l.startswith("{")][-1])
    h = d.get("roofline_hbm", {})
    print(f"{v:10s} step {d['ms_per_step']:.2f} ms  gru fwd/bwd/wgrad " + " / ".join(f"{h[k]['ms_per_step']:.3f}" for k in ("gru_fwd", "gru_bwd", "gru_wgrad")), flush=True)
except Exception as e:
    print(v, "FAILED", e, open(f"gpurun_out/gru_ab_{v}.err").read()[-1500:])
PY
done
unset DF_LIB
