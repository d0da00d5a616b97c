#!/bin/bash
# real victim (SLP-built pillar feature net backward) beside SYNTHETIC neighbours: a register-resident MFMA stream per instruction
cd ${GRAFT_REPO_ROOT:-.}
SLP=$PWD/deflow_amd/_build/deflow_amd_slp/libdeflow_amd_slp.so
run() { "$@" 2>&1 | grep -E "pfn backward|Error|error" | cut -c1-260; }
for k in 0 1 2 3; do
  echo "== slp build, side stream mfma:$k"
  DF_LIB=$SLP DF_STRESS_SIDE=mfma:$k run python tools/pfn_bwd_stress.py 8000
done
