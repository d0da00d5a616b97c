#!/bin/bash
# Is the cross-process failure of the SLP-built pillar feature net backward an UNINITIALISED READ?  Neighbour = tools/poison.hip
# (NaN in 240 vector registers and all 160 KB of LDS of every CU, over and over), as a second process and as a side stream.
cd ${GRAFT_REPO_ROOT:-.}
SLP=$PWD/deflow_amd/_build/deflow_amd_slp/libdeflow_amd_slp.so
REPS=${REPS:-20000}
run() { "$@" 2>&1 | grep -E "pfn backward|reps differ|Error|error" | cut -c1-260; }
for what in 7 1 2; do
  echo "== slp build, poison($what) second process"
  tools/bin/poison 600 $what > /tmp/poison.log 2>&1 &
  NB=$!
  sleep 3
  DF_LIB=$SLP run python tools/pfn_bwd_stress.py $REPS
  kill $NB 2>/dev/null; wait $NB 2>/dev/null
done
echo "== slp build, poison(7) side stream, same process"
DF_LIB=$SLP DF_STRESS_SIDE=poison:7 run python tools/pfn_bwd_stress.py $((REPS / 2))
echo "== default build, poison(7) second process"
tools/bin/poison 600 7 > /tmp/poison.log 2>&1 &
NB=$!
sleep 3
run python tools/pfn_bwd_stress.py $REPS
kill $NB 2>/dev/null; wait $NB 2>/dev/null
echo "== default build, poison(7) side stream"
DF_STRESS_SIDE=poison:7 run python tools/pfn_bwd_stress.py $((REPS / 2))
