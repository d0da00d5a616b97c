"""End-point-error metrics on ``pose_flow[valid] + flow`` (the reference's eval.py and
metric code live in the absent OpenSceneFlow submodule -- these are the standard scene-flow definitions: EPE,
strict/relaxed accuracy, and the 3-way split used by the Argoverse-2 leaderboard with a dynamic mask of
|gt - ego| > 0.05 m/frame).  Host-side bookkeeping on small per-sample tensors, not part of the hot path."""
from __future__ import annotations

from typing import Dict, Optional

import torch


def epe_metrics(est_flow: torch.Tensor, gt_flow: torch.Tensor, pose_flow: Optional[torch.Tensor] = None,
                foreground: Optional[torch.Tensor] = None) -> Dict[str, float]:
    """est_flow / gt_flow [M,3] TOTAL flow (ego motion included); pose_flow [M,3] ego-motion flow of the same points."""
    ok = torch.isfinite(est_flow).all(1) & torch.isfinite(gt_flow).all(1)
    est, gt = est_flow[ok], gt_flow[ok]
    err = (est - gt).norm(dim=1)
    gtn = gt.norm(dim=1)
    rel = err / gtn.clamp_min(1e-9)
    out = {"EPE": float(err.mean()) if err.numel() else float("nan"),
           "AccS": float(((err < 0.05) | (rel < 0.05)).float().mean()) if err.numel() else float("nan"),
           "AccR": float(((err < 0.10) | (rel < 0.10)).float().mean()) if err.numel() else float("nan"),
           "n": int(err.numel())}
    if pose_flow is not None:
        dyn = (gt - pose_flow[ok]).norm(dim=1) > 0.05
        fg = foreground[ok] if foreground is not None else torch.ones_like(dyn)

        def m(sel):
            return float(err[sel].mean()) if bool(sel.any()) else float("nan")
        out.update({"EPE_FD": m(fg & dyn), "EPE_FS": m(fg & ~dyn), "EPE_BS": m(~fg & ~dyn)})
        vals = [v for v in (out["EPE_FD"], out["EPE_FS"], out["EPE_BS"]) if v == v]
        out["EPE_3way"] = sum(vals) / len(vals) if vals else float("nan")
    return out


def evaluate_batch(res: dict, batch: dict) -> Dict[str, float]:
    """Average the metrics of one model(batch) result dict over its samples (final flow = pose_flow[valid] + flow).
    With labelled scene files the batch also carries ``flow_is_valid`` (points without a usable label are left out) and
    ``flow_category_indices`` (0 = background / no object, anything else foreground -- the Argoverse-2 split)."""
    acc: Dict[str, list] = {}
    for b in range(len(res["flow"])):
        vi = res["pc0_valid_point_idxes"][b]
        pf = res["pose_flow"][b][vi]
        est, gt = pf + res["flow"][b].detach(), batch["flow"][b][vi]
        fg = None
        if "flow_category_indices" in batch:
            fg = batch["flow_category_indices"][b][vi] != 0
        if "flow_is_valid" in batch:
            ok = batch["flow_is_valid"][b][vi].bool()
            est, gt, pf = est[ok], gt[ok], pf[ok]
            fg = fg[ok] if fg is not None else None
        m = epe_metrics(est, gt, pf, fg)
        for k, v in m.items():
            if v == v:
                acc.setdefault(k, []).append(v)
    return {k: sum(v) / len(v) for k, v in acc.items()}
