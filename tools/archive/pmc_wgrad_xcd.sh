cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for X in 0 1; do
  DF_WGRAD_XCD=$X rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/fx$X -o p -- python $R/tools/ab_wgrad_bf16.py > /tmp/fx$X.log 2>&1
  echo "XCD=$X"; python $R/tools/rocpd_pmc.py --raw wgrad3_ring $(find /tmp/fx$X -name "*.db" | head -1)
done
