"""The two ablation losses of the reference's ``loss_fn=`` switch next to deflowLoss ([REF assets/slurm/1_train.sh:58-78]:
``loss_fn = [ff3dLoss (R), zeroflowLoss, deflowLoss]``; [REF README.md:68]: the fastflow3d baseline trains with ff3dLoss).
Their definitions live in the absent OpenSceneFlow submodule (``src/lossfuncs.py``) and are recalled from FastFlow3D / ZeroFlow:

  ff3dLoss      mean over points of |est - gt| * (0.1 for background points (class 0), 1.0 for foreground)
  zeroflowLoss  mean over points of |est - gt| * clamp(1.8 * speed - 0.8, 0.1, 1.0),  speed = |gt| * 10 (m/s at 10 Hz)

each per sample, summed over the batch like deflowLoss in the trainer.  deflowLoss -- the north-star loss -- has its own
HIP kernels (autograd.DeflowLossFn).  Round 5: the Trainer's direct step evaluates these two with HIP kernels as well
(csrc/misc.hip: df_wloss_fwd / _finalize / _bwd, the same definitions); the torch form below serves autograd callers
(Trainer.loss_on_last_forward) and is the kernels' twin in tests/test_gpu_model.py::test_ablation_losses_vs_oracle."""
from __future__ import annotations

from typing import Optional

import torch


def _rows(est: torch.Tensor, gt: torch.Tensor, counts: torch.Tensor):
    B, N, _ = est.shape
    valid = torch.arange(N, device=est.device)[None, :] < counts[:, None]
    valid = valid & torch.isfinite(gt).all(-1) & torch.isfinite(est.detach()).all(-1)
    diff = torch.where(valid[..., None], est - gt, torch.zeros_like(est))
    # |d| with a zero (not NaN) gradient at masked rows
    err = torch.where(valid, torch.linalg.vector_norm(torch.where(valid[..., None], diff, torch.ones_like(diff)), dim=-1),
                      torch.zeros_like(diff[..., 0]))
    return err, valid


def _sum_of_sample_means(err: torch.Tensor, weight: torch.Tensor, valid: torch.Tensor) -> torch.Tensor:
    n = valid.sum(1)
    per = (err * weight * valid).sum(1) / n.clamp_min(1)
    return per[n > 0].sum()


def ff3d_loss(est: torch.Tensor, gt: torch.Tensor, counts: torch.Tensor, classes: Optional[torch.Tensor]) -> torch.Tensor:
    if classes is None:
        raise ValueError("loss_fn=ff3dLoss needs batch['flow_category_indices'] (labelled scene files)")
    err, valid = _rows(est, gt, counts)
    return _sum_of_sample_means(err, (classes > 0).float() * 0.9 + 0.1, valid)


def zeroflow_loss(est: torch.Tensor, gt: torch.Tensor, counts: torch.Tensor) -> torch.Tensor:
    err, valid = _rows(est, gt, counts)
    speed = torch.linalg.vector_norm(torch.where(valid[..., None], gt, torch.zeros_like(gt)), dim=-1) * 10.0
    return _sum_of_sample_means(err, torch.clamp(1.8 * speed - 0.8, 0.1, 1.0), valid)
