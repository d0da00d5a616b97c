"""experiment (DF_EXP_PROF build of conv.hip): cycles a persistent 3x3 workgroup spends in the K loop vs in the epilogue, per tile,
summed over three training steps.  DF_LIB=deflow_amd/_build/exp/libPROF.so python tools/exp_x3p_prof.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deflow_amd
from deflow_amd import _lib
from deflow_amd.optim import Trainer
from deflow_amd.synth import synth_batch
dev = torch.device("cuda")
model = deflow_amd.DeFlow(grid_feature_size=[512, 512], num_iters=4).to(dev).train()
tr = Trainer(model, lr=2e-4)
bs = [synth_batch(16, 80000, seed=20240116 + i, device=dev) for i in range(2)]
for i in range(3):
    tr.step(bs[i % 2])
lib = ctypes.CDLL(_lib.LIB_PATH)
out = (ctypes.c_ulonglong * 8)()
assert lib.df_exp_prof(out, 1) == 0
for i in range(3):
    tr.step(bs[i % 2])
assert lib.df_exp_prof(out, 0) == 0
for name, o in (("BN=64 tiles", 0), ("BN=128 tiles", 4)):
    m, e, n = out[o], out[o + 1], out[o + 2]
    if n:
        print(f"{name}: {n} tiles, K loop {m / n:.0f} cycles/tile, epilogue {e / n:.0f} cycles/tile = {100 * e / (m + e):.1f} % of the workgroup's time")
