#!/bin/bash
# Does the failure need the two processes to run DIFFERENT code for the pillar kernels (an instruction-cache / address-aliasing effect
# between the processes' code objects), or does it appear with identical libraries too?
cd ${GRAFT_REPO_ROOT:-.}
SLP=$PWD/deflow_amd/_build/deflow_amd_slp/libdeflow_amd_slp.so
REPS=${REPS:-20000}
run() { "$@" 2>&1 | grep -E "pfn backward|Error|error" | cut -c1-260; }
pair() {  # victim lib, neighbour lib
  echo "== victim $(basename ${1:-default}), neighbour (fp32 training step) $(basename ${2:-default})"
  DF_LIB=$2 python tools/pfn_neighbour.py train_fp32 600 > /tmp/nb.log 2>&1 &
  NB=$!
  sleep 12
  DF_LIB=$1 run python tools/pfn_bwd_stress.py $REPS
  kill $NB 2>/dev/null; wait $NB 2>/dev/null
}
pair $SLP $SLP
pair $SLP ""
pair "" $SLP
pair "" ""
echo "== poison check"; tools/bin/poison 3 7
