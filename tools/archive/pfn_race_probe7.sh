#!/bin/bash
# which kernel of the inference forward does the SLP-built victim react to?  neighbour THREAD = infer with only some entry points live
cd ${GRAFT_REPO_ROOT:-.}
SLP=$PWD/deflow_amd/_build/deflow_amd_slp/libdeflow_amd_slp.so
REPS=${REPS:-15000}
run() { "$@" 2>&1 | grep -E "pfn backward|neighbour thread|Error|error" | cut -c1-260; }
for only in "${@}"; do
  echo "== neighbour thread: infer, only /$only/"
  DF_NB_ONLY="$only" DF_LIB=$SLP DF_STRESS_THREAD=infer run python tools/pfn_bwd_stress.py $REPS
done
