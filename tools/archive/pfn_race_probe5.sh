#!/bin/bash
# bisection of the neighbour PROCESS by what it runs (victim: SLP-built pillar feature net backward)
cd ${GRAFT_REPO_ROOT:-.}
SLP=$PWD/deflow_amd/_build/deflow_amd_slp/libdeflow_amd_slp.so
REPS=${REPS:-20000}
run() { "$@" 2>&1 | grep -E "pfn backward|Error|error" | cut -c1-260; }
for kind in ${KINDS:-torch_ops embed infer fwd_train train_linear}; do
  echo "== neighbour process: $kind"
  python tools/pfn_neighbour.py $kind 600 > /tmp/nb_$kind.log 2>&1 &
  NB=$!
  sleep 12
  DF_LIB=$SLP run python tools/pfn_bwd_stress.py $REPS
  kill $NB 2>/dev/null; wait $NB 2>/dev/null
  grep -i -E "error|Traceback" /tmp/nb_$kind.log | head -3
done
