# HBM bytes per launch of every kernel of the training step: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only),
# gfx950 correction in tools/rocpd_pmc.py.  usage: bash tools/pmc_hbm.sh TAG  ->  gpurun_out/TAG_pmc_hbm_bytes.txt
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o p -- python $R/bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline --no-loader --no-profile > /tmp/pmc_$c.log 2>&1
done
F=$(find /tmp/pmc_FETCH_SIZE -name "*.db" | head -1); W=$(find /tmp/pmc_WRITE_SIZE -name "*.db" | head -1)
python $R/tools/rocpd_pmc.py $F $W > $R/gpurun_out/${TAG}_pmc_hbm_bytes.txt 2>&1
head -${HEADN:-45} $R/gpurun_out/${TAG}_pmc_hbm_bytes.txt
