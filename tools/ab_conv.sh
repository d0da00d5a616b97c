for rep in 1 2 3; do for d in 0 1; do echo -n "WGRAD_W8=$d "; DF_WGRAD_W8=$d python tools/ab_wgrad.py 2>&1 | grep wgrad; done; done
