#!/bin/bash
# the neighbour as a second host THREAD of the victim's process (own stream): does the failure need a second PROCESS?
cd ${GRAFT_REPO_ROOT:-.}
SLP=$PWD/deflow_amd/_build/deflow_amd_slp/libdeflow_amd_slp.so
REPS=${REPS:-20000}
run() { "$@" 2>&1 | grep -E "pfn backward|neighbour thread|Error|error" | cut -c1-260; }
for kind in infer train_fp32; do
  echo "== slp build, neighbour THREAD: $kind"
  DF_LIB=$SLP DF_STRESS_THREAD=$kind run python tools/pfn_bwd_stress.py $REPS
done
echo "== slp build, neighbour PROCESS: infer (control)"
python tools/pfn_neighbour.py infer 600 > /tmp/nb.log 2>&1 &
NB=$!
sleep 12
DF_LIB=$SLP run python tools/pfn_bwd_stress.py $REPS
kill $NB 2>/dev/null; wait $NB 2>/dev/null
