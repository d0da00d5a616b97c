#!/bin/bash
# Is the WHOLE small training step bit-reproducible beside a neighbour PROCESS that runs only the GRU forward of the inference pass?
# default library (pillarize.hip without SLP, every other file with) vs a library built entirely with -fno-slp-vectorize.
# (An in-process neighbour THREAD cannot drive deflow_amd.ops beside a Trainer: the module state -- bound slots, weight forms -- is per
# process, not per thread.)
cd ${GRAFT_REPO_ROOT:-.}
NOSLP=$PWD/deflow_amd/_build/deflow_amd_noslp/libdeflow_amd_noslp.so
run() { "$@" 2>&1 | grep -E "varying gradient|repetitions differ|Error|error" | cut -c1-220 | head -8; }
DF_NB_ONLY=df_gru python tools/pfn_neighbour.py infer 900 > /tmp/nb.log 2>&1 &
NB=$!
sleep 12
for dt in fp32 bf16; do
  echo "== default library, $dt step, neighbour process infer /df_gru/"
  run python tools/grad_repro_probe.py 3000 $dt
  echo "== all-noslp library, $dt step, neighbour process infer /df_gru/"
  DF_LIB=$NOSLP run python tools/grad_repro_probe.py 3000 $dt
done
kill $NB 2>/dev/null; wait $NB 2>/dev/null
echo "== default library, fp32 step, no neighbour"
run python tools/grad_repro_probe.py 1000 fp32
