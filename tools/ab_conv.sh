for rep in 1 2 3; do for d in 0 3 2; do echo -n "RING=$d "; DF_WGRAD_RING=$d python tools/ab_wgrad.py 2>&1 | grep wgrad; done; done
