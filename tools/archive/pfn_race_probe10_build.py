"""builds the library variants of tools/pfn_race_probe10.sh (CPU, hipcc cross-compiles)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deflow_amd import build as B
for name, flags in (("slp", ["-fslp-vectorize"]), ("slp_lb1", ["-fslp-vectorize", "-DDF_GRU_LB=1"]), ("slp_v128", ["-fslp-vectorize", "-DDF_GRU_NUM_VGPR=128"])):
    print(B.build_variant(name, flags))
