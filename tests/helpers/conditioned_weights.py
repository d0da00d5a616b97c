"""Weights "as a run would have them" for the bf16 gradient check: 50 fp32 Adam steps of the HIP trainer from the
seeded initialisation, on five rotating synthetic batches of a SMALL grid (256 x 256, 20 000 points, 4 pairs, 8 GRU iterations --
the parameters do not depend on the grid, so the state loads into the configs[4]-shaped model), BatchNorm running statistics
included.  The engine is deterministic, so the GPU box regenerates the same state the float64 digest was computed for
(oracle/gen_digest_bs16.py --weights; a later library whose kernels round differently reproduces it to ~1e-6, far inside the
bound the test states).  Used by tests/test_gpu_model.py::test_bf16_gradients_on_conditioned_weights and by
`python tests/helpers/conditioned_weights.py out.pt` (the dump the digest was generated from)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

STEPS, INIT_SEED, DATA_SEED = 50, 46, 9100


def conditioned_state(dev, steps: int = STEPS):
    import deflow_amd
    from deflow_amd.optim import Trainer
    from deflow_amd.synth import synth_batch
    from oracle import ref_torch as O     # (only for the seeded initial weights, as the other digest tests take them)
    cfg = dict(voxel_size=[0.2, 0.2, 6], point_cloud_range=[-25.6, -25.6, -3, 25.6, 25.6, 3], grid_feature_size=[256, 256], num_iters=8)
    torch.manual_seed(INIT_SEED)
    ref = O.DeFlow(**cfg)
    m = deflow_amd.DeFlow(**cfg)
    m.load_state_dict(ref.state_dict())
    m = m.to(dev).train()
    tr = Trainer(m, lr=2e-4)
    batches = [{k: v.to(dev) for k, v in synth_batch(4, 20000, seed=DATA_SEED + 17 * j, grid_hw=(256, 256), exact=True).items()} for j in range(5)]
    losses = []
    for i in range(steps):
        losses.append(tr.step(batches[i % len(batches)]))
    torch.cuda.synchronize()
    return {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}, [float(l) for l in losses]


if __name__ == "__main__":
    sd, losses = conditioned_state(torch.device("cuda", 0))
    torch.save(sd, sys.argv[1])
    print("losses", " ".join(f"{l:.4f}" for l in losses[::7]), "->", sys.argv[1], os.path.getsize(sys.argv[1]) >> 20, "MiB")
