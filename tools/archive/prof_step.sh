# Refresh the round's profile artefacts on the GPU box (run through gpurun; results land in gpurun_out/).
#   1. rocprofv3 --kernel-trace --stats of the default bench command -> kernel table (tools/rocpd_stats.py)
#   2. two separate PMC passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only, as the pool requires) -> HBM bytes per launch
TAG=${1:-r01_g}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py > /tmp/kt.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/kt -name "*.db" | head -1) > $R/gpurun_out/${TAG}_kernel_stats.txt 2>&1
grep "^{\"metric\"" /tmp/kt.log | tail -1 > $R/gpurun_out/${TAG}_bench.json
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o p -- python $R/bench.py --steps 1 --warmup 1 > /tmp/pmc_$c.log 2>&1
done
F=$(find /tmp/pmc_FETCH_SIZE -name "*.db" | head -1); W=$(find /tmp/pmc_WRITE_SIZE -name "*.db" | head -1)
python $R/tools/rocpd_pmc.py $F $W > $R/gpurun_out/${TAG}_pmc_hbm_bytes.txt 2>&1
python $R/tools/rocpd_pmc.py --json $R/gpurun_out/pmc_traffic.json $F $W
head -12 $R/gpurun_out/${TAG}_kernel_stats.txt; head -8 $R/gpurun_out/${TAG}_pmc_hbm_bytes.txt; cat $R/gpurun_out/${TAG}_bench.json | cut -c1-300
