cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
for rot in 1 0; do
  DF_CONV_ROT=$rot rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmcF_$rot -o p -- python $R/bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline > /tmp/pmcF_$rot.log 2>&1
  F=$(find /tmp/pmcF_$rot -name "*.db" | head -1)
  python $R/tools/rocpd_pmc.py --raw conv_halo_x3 $F > $O/pmc_rot$rot.txt 2>&1 || tail -3 /tmp/pmcF_$rot.log >> $O/pmc_rot$rot.txt
  python $R/tools/rocpd_pmc.py --raw wgrad3_h2p $F >> $O/pmc_rot$rot.txt 2>&1
done
