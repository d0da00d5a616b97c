#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python tools/canary_probe.py none 2000 8192 2>&1 | grep canary
DF_NB_ONLY=df_gru python tools/canary_probe.py infer 6000 8192 2>&1 | grep -E "canary|Error"
DF_NB_ONLY=df_gru python tools/canary_probe.py infer 6000 36864 2>&1 | grep -E "canary|Error"
python tools/canary_probe.py infer 6000 8192 2>&1 | grep -E "canary|Error"
