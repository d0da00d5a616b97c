cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmcW -o p -- python $R/bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline > /tmp/pmcW.log 2>&1
F=$(find /tmp/pmcW -name "*.db" | head -1)
python $R/tools/rocpd_pmc.py --raw wgrad3_h2p $F > $O/pmc_wgrad.txt 2>&1 || tail -3 /tmp/pmcW.log >> $O/pmc_wgrad.txt
