#!/bin/bash
# round 5: does the OCCUPANCY of the neighbour move the failure rate of the SLP-built pillar feature net backward?
# victim: tools/pfn_bwd_stress.py on a library whose pillarize.hip is SLP-vectorised; neighbour: a host thread running only the GRU
# decoder's forward (DF_NB_ONLY=df_gru), generation 3 (the original trigger, DF_GRU_LEAN=0) and generation 4 (the lean kernel), each built
#   slp        as shipped (launch bounds (256, 2): two workgroups per CU)
#   slp_lb1    -DDF_GRU_LB=1            one workgroup per CU may use up to 512 registers per lane
#   slp_v128   -DDF_GRU_NUM_VGPR=128    a 128-register budget (spills; more waves fit beside it)
# build the variants first (CPU): python tools/pfn_race_probe10_build.py
cd ${GRAFT_REPO_ROOT:-.}
REPS=${REPS:-6000}
run() { "$@" 2>&1 | grep -E "pfn backward|neighbour thread|Error|error" | cut -c1-220; }
for v in slp slp_lb1 slp_v128; do
  LIB=$PWD/deflow_amd/_build/$v/lib$v.so
  [ -f $LIB ] || { echo "missing $LIB"; continue; }
  for lean in 0 1; do
    echo "== library $v, neighbour thread = GRU forward generation $((3 + lean))"
    DF_LIB=$LIB DF_GRU_LEAN=$lean DF_STRESS_THREAD=infer DF_NB_ONLY=df_gru run python tools/pfn_bwd_stress.py $REPS
  done
done
echo "== shipped library (pillarize.hip without the SLP vectoriser), neighbour generation 3"
DF_GRU_LEAN=0 DF_STRESS_THREAD=infer DF_NB_ONLY=df_gru run python tools/pfn_bwd_stress.py $REPS
