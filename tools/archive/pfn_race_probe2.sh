#!/bin/bash
# bisection of the neighbour (see pfn_race_probe.sh): SLP-built victim; neighbour kind x {second process, side stream of the same process}
cd ${GRAFT_REPO_ROOT:-.}
SLP=$PWD/deflow_amd/_build/deflow_amd_slp/libdeflow_amd_slp.so
REPS=${REPS:-20000}
run() { "$@" 2>&1 | grep -E "pfn backward|Error|error" | cut -c1-200; }
for kind in ${KINDS:-train_bf16 train_fp32 conv_bf16 conv_h2 conv_fp32 wgrad_bf16 wgrad_h2 bn_gelu}; do
  echo "== neighbour $kind: second process"
  python tools/pfn_neighbour.py $kind 600 > /tmp/nb_$kind.log 2>&1 &
  NB=$!
  sleep 12
  DF_LIB=$SLP run python tools/pfn_bwd_stress.py $REPS
  kill $NB 2>/dev/null; wait $NB 2>/dev/null
  tail -n 2 /tmp/nb_$kind.log | grep -i -E "error|Traceback" 
  echo "== neighbour $kind: side stream, same process"
  DF_LIB=$SLP DF_STRESS_SIDE=$kind run python tools/pfn_bwd_stress.py $((REPS / 4))
done
