"""Evaluation metrics of ``eval.py ... av2_mode=val`` on ``pose_flow[valid] + flow`` (SURVEY.md section 8(f) row N3;
[REF README.md:88-91; assets/slurm/2_eval.sh:33-35] "it will directly prints all metric").

The reference's metric code lives in the absent OpenSceneFlow submodule (``src/utils/eval_metric.py`` / ``av2_eval.py``, following
av2-api ``av2/evaluation/scene_flow/eval.py``); these are the published Argoverse-2 definitions:

  leaderboard_version=1   three-way EPE: mean end-point error of Foreground-Dynamic / Foreground-Static / Background-Static points
                          inside the 70 m x 70 m box around the sensor (|x|, |y| <= 35 m), dynamic = |gt - ego| >= 0.05 m per frame,
                          foreground = any annotated category (label != 0); "Three-way" = mean of the three; dynamic IoU; plus EPE,
                          strict / relaxed accuracy (error < 0.05 / 0.10 m OR relative error < 5 / 10 %) and the space-time angle error
                          over the valid points of the box.  Per-frame values, averaged over frames.
  leaderboard_version=2   bucketed normalised EPE: points within 35 m (xy, Euclidean), ego motion removed, five meta-classes x 51 speed
                          buckets (0.04 m per frame wide up to 2.0, then open); per class the static EPE (first bucket) and the mean over
                          the non-empty dynamic buckets of EPE / mean speed; their means over classes.  Count-weighted over frames.

Host-side bookkeeping on per-sample tensors (a few vectorised torch ops per frame, on whatever device the flow lives), not part of
the hot path.  tests/ check every number against the numpy restatement in oracle/ref_metrics.py."""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

CLOSE_DISTANCE_THRESHOLD = 35.0
DYNAMIC_THRESHOLD = 0.05       # m per frame between two 10 Hz sweeps = 0.5 m/s
N_CATEGORIES = 31              # 0 = no annotation, 1..30 = av2 AnnotationCategories in alphabetical order
META_CLASSES = ("BACKGROUND", "CAR", "OTHER_VEHICLES", "PEDESTRIAN", "WHEELED_VRU")
# label index -> meta-class (-1: not evaluated by the bucketed metric, e.g. signs / bollards / animals)
_META_OF = [-1] * N_CATEGORIES
_META_OF[0] = 0
for _i in (19,):                                   # REGULAR_VEHICLE
    _META_OF[_i] = 1
for _i in (6, 11, 18, 25, 26, 27, 2, 7, 20):       # BOX_TRUCK LARGE_VEHICLE RAILED_VEHICLE TRUCK TRUCK_CAB VEHICULAR_TRAILER ARTICULATED_BUS BUS SCHOOL_BUS
    _META_OF[_i] = 2
for _i in (17, 23, 28, 16):                        # PEDESTRIAN STROLLER WHEELCHAIR OFFICIAL_SIGNALER
    _META_OF[_i] = 3
for _i in (3, 4, 14, 15, 29, 30):                  # BICYCLE BICYCLIST MOTORCYCLE MOTORCYCLIST WHEELED_DEVICE WHEELED_RIDER
    _META_OF[_i] = 4
N_BUCKETS = 51                 # [0.04 k, 0.04 (k + 1)) for k < 50, then [2.0, inf)
BUCKET_WIDTH = 2.0 / 50


def _acc(err: torch.Tensor, gtn: torch.Tensor, thr: float) -> torch.Tensor:
    return ((err < thr) | (err / (gtn + 1e-10) < thr)).double()


def _angle(est: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    """angle between the space-time vectors (flow, 0.1)"""
    dot = (est * gt).sum(1) + 0.01
    n = torch.sqrt((est * est).sum(1) + 0.01) * torch.sqrt((gt * gt).sum(1) + 0.01)
    return torch.arccos((dot / n).clamp(-1.0, 1.0))


def epe_metrics(est_flow: torch.Tensor, gt_flow: torch.Tensor, pose_flow: Optional[torch.Tensor] = None,
                foreground: Optional[torch.Tensor] = None) -> Dict[str, float]:
    """Range-free summary of one sample (the trainer's per-epoch validation line): est_flow / gt_flow [M,3] TOTAL flow (ego motion
    included); pose_flow [M,3] ego-motion flow of the same points.  For the leaderboard tables use OfficialMetrics."""
    ok = torch.isfinite(est_flow).all(1) & torch.isfinite(gt_flow).all(1)
    est, gt = est_flow[ok], gt_flow[ok]
    err = (est - gt).norm(dim=1)
    gtn = gt.norm(dim=1)
    out = {"EPE": float(err.mean()) if err.numel() else float("nan"),
           "AccS": float(_acc(err, gtn, 0.05).mean()) if err.numel() else float("nan"),
           "AccR": float(_acc(err, gtn, 0.10).mean()) if err.numel() else float("nan"),
           "n": int(err.numel())}
    if pose_flow is not None:
        dyn = (gt - pose_flow[ok]).norm(dim=1) >= DYNAMIC_THRESHOLD
        fg = foreground[ok] if foreground is not None else torch.ones_like(dyn)

        def m(sel):
            return float(err[sel].mean()) if bool(sel.any()) else float("nan")
        out.update({"EPE_FD": m(fg & dyn), "EPE_FS": m(fg & ~dyn), "EPE_BS": m(~fg & ~dyn)})
        vals = [v for v in (out["EPE_FD"], out["EPE_FS"], out["EPE_BS"]) if v == v]
        out["EPE_3way"] = sum(vals) / len(vals) if vals else float("nan")
    return out


class OfficialMetrics:
    """Accumulates the Argoverse-2 validation tables over frames: ``step`` once per sample, ``result(leaderboard_version)`` at the end."""

    V1_KEYS = ("EPE_FD", "EPE_FS", "EPE_BS", "IoU", "EPE", "AccS", "AccR", "Angle")

    def __init__(self):
        self.v1_sum = {k: 0.0 for k in self.V1_KEYS}
        self.v1_cnt = {k: 0 for k in self.V1_KEYS}
        self.n = 0
        self.err_sum = torch.zeros(len(META_CLASSES), N_BUCKETS, dtype=torch.float64)
        self.speed_sum = torch.zeros(len(META_CLASSES), N_BUCKETS, dtype=torch.float64)
        self.count = torch.zeros(len(META_CLASSES), N_BUCKETS, dtype=torch.int64)

    @torch.no_grad()
    def step(self, est_flow: torch.Tensor, rigid_flow: torch.Tensor, pc0: torch.Tensor, gt_flow: torch.Tensor,
             is_valid: Optional[torch.Tensor] = None, categories: Optional[torch.Tensor] = None) -> None:
        """one frame: total estimated / ego / ground-truth flow [M,3] of the points pc0 [M,3] (sensor frame of the first sweep),
        is_valid [M] (points with a usable label; default all), categories [M] (av2 label index, 0 = none; default all 0)"""
        M = est_flow.shape[0]
        dev = est_flow.device
        est, rigid, gt, pc = (t.double() for t in (est_flow, rigid_flow, gt_flow, pc0[:, :3]))
        valid = torch.ones(M, dtype=torch.bool, device=dev) if is_valid is None else is_valid.bool()
        cats = torch.zeros(M, dtype=torch.long, device=dev) if categories is None else categories.long().clamp(0, N_CATEGORIES - 1)
        finite = torch.isfinite(est).all(1) & torch.isfinite(rigid).all(1) & torch.isfinite(gt).all(1) & torch.isfinite(pc).all(1)
        err = (est - gt).norm(dim=1)
        # ---- version 1: box |x|, |y| <= 35 m, three-way split, segmentation IoU ----
        box = (pc[:, :2].abs() <= CLOSE_DISTANCE_THRESHOLD).all(1)
        sel = finite & valid & box
        gt_dyn = (gt - rigid).norm(dim=1) >= DYNAMIC_THRESHOLD
        est_dyn = (est - rigid).norm(dim=1) >= DYNAMIC_THRESHOLD
        fg = cats != 0

        def put(key: str, mask: torch.Tensor, values: torch.Tensor):
            c = int(mask.sum())
            if c:
                self.v1_sum[key] += float(values[mask].sum()) / c
                self.v1_cnt[key] += 1
        put("EPE_FD", sel & fg & gt_dyn, err)
        put("EPE_FS", sel & fg & ~gt_dyn, err)
        put("EPE_BS", sel & ~fg & ~gt_dyn, err)
        tp = int((sel & est_dyn & gt_dyn).sum())
        fp = int((sel & est_dyn & ~gt_dyn).sum())
        fn = int((sel & ~est_dyn & gt_dyn).sum())
        if tp + fp + fn:
            self.v1_sum["IoU"] += tp / (tp + fp + fn)
            self.v1_cnt["IoU"] += 1
        gtn = gt.norm(dim=1)
        put("EPE", sel, err)
        put("AccS", sel, _acc(err, gtn, 0.05))
        put("AccR", sel, _acc(err, gtn, 0.10))
        put("Angle", sel, _angle(torch.where(sel[:, None], est, torch.zeros_like(est)), torch.where(sel[:, None], gt, torch.zeros_like(gt))))
        self.n += int(sel.sum())
        # ---- version 2: 35 m radius, ego motion removed, meta-class x speed-bucket sums ----
        meta = torch.tensor(_META_OF, device=dev)[cats]
        sel2 = finite & valid & (pc[:, :2].norm(dim=1) <= CLOSE_DISTANCE_THRESHOLD) & (meta >= 0)
        if bool(sel2.any()):
            speed = (gt - rigid).norm(dim=1)[sel2]
            e2 = err[sel2]            # |(est - rigid) - (gt - rigid)| = |est - gt|
            edges = torch.arange(1, N_BUCKETS, device=dev, dtype=torch.float64) * BUCKET_WIDTH     # 0.04 .. 2.0
            bucket = torch.bucketize(speed, edges, right=True)    # speed in [edge[k-1], edge[k]) -> k
            cell = meta[sel2] * N_BUCKETS + bucket
            nc = len(META_CLASSES) * N_BUCKETS
            self.err_sum += torch.zeros(nc, dtype=torch.float64, device=dev).index_add_(0, cell, e2).view(-1, N_BUCKETS).cpu()
            self.speed_sum += torch.zeros(nc, dtype=torch.float64, device=dev).index_add_(0, cell, speed).view(-1, N_BUCKETS).cpu()
            self.count += torch.bincount(cell, minlength=nc).view(-1, N_BUCKETS).cpu()

    def result(self, leaderboard_version: int = 1) -> Dict[str, float]:
        if int(leaderboard_version) == 1:
            out = {k: (self.v1_sum[k] / self.v1_cnt[k] if self.v1_cnt[k] else float("nan")) for k in self.V1_KEYS}
            three = [out[k] for k in ("EPE_FD", "EPE_FS", "EPE_BS")]
            out["Three-way"] = sum(three) / 3 if not any(math.isnan(v) for v in three) else float("nan")
            out["n"] = self.n
            return out
        out: Dict[str, float] = {}
        stat, dyn = [], []
        for ci, name in enumerate(META_CLASSES):
            cnt = self.count[ci]
            s = float(self.err_sum[ci, 0] / cnt[0]) if cnt[0] > 0 else float("nan")
            nz = cnt[1:] > 0
            # (EPE mean) / (speed mean) of a bucket = err_sum / speed_sum
            d = float((self.err_sum[ci, 1:][nz] / self.speed_sum[ci, 1:][nz]).mean()) if bool(nz.any()) else float("nan")
            out[f"{name}/Static"], out[f"{name}/Dynamic"] = s, d
            stat.append(s)
            dyn.append(d)
        for key, vals in (("mean/Static", stat), ("mean/Dynamic", dyn)):
            vals = [v for v in vals if not math.isnan(v)]
            out[key] = sum(vals) / len(vals) if vals else float("nan")
        return out

    def table(self, leaderboard_version: int = 1) -> str:
        r = self.result(leaderboard_version)
        if int(leaderboard_version) == 1:
            head = ["Three-way", "EPE_FD", "EPE_FS", "EPE_BS", "IoU", "EPE", "AccS", "AccR", "Angle"]
            return " | ".join(f"{k} {r[k]:.4f}" for k in head) + f" | n {r['n']}"
        rows = [f"{'class':16s} {'Static':>10s} {'Dynamic':>10s}"]
        for name in META_CLASSES + ("mean",):
            rows.append(f"{name:16s} {r[name + '/Static']:10.4f} {r[name + '/Dynamic']:10.4f}")
        return "\n".join(rows)


def evaluate_batch(res: dict, batch: dict, official: Optional[OfficialMetrics] = None) -> Dict[str, float]:
    """Average the range-free summary over the samples of one model(batch) result dict (final flow = pose_flow[valid] + flow), and
    feed ``official`` (the leaderboard tables) frame by frame when given.  With labelled scene files the batch also carries
    ``flow_is_valid`` (points without a usable label are left out), ``flow_category_indices`` (0 = no annotation) and, for the
    official validation split, ``eval_mask`` (the benchmark's point mask)."""
    acc: Dict[str, list] = {}
    # frames WITHOUT the benchmark's mask in a batch where other frames carry one (``has_eval_mask`` from the collate function) are not
    # official evaluation frames: they are left out, as an evaluation over upstream's index_eval.pkl never sees them
    has = batch.get("has_eval_mask")
    skip = (~has.bool()).tolist() if has is not None and bool(has.any()) else None
    for b in range(len(res["flow"])):
        if skip is not None and skip[b]:
            continue
        vi = res["pc0_valid_point_idxes"][b]
        pf = res["pose_flow"][b][vi]
        est, gt = pf + res["flow"][b].detach(), batch["flow"][b][vi]
        cats = batch["flow_category_indices"][b][vi] if "flow_category_indices" in batch else None
        ok = batch["flow_is_valid"][b][vi].bool() if "flow_is_valid" in batch else None
        if "eval_mask" in batch:
            em = batch["eval_mask"][b][vi].bool()
            ok = em if ok is None else ok & em
        if official is not None:
            official.step(est, pf, batch["pc0"][b][vi], gt, ok, cats)
        fg = cats != 0 if cats is not None else None
        if ok is not None:
            est, gt, pf = est[ok], gt[ok], pf[ok]
            fg = fg[ok] if fg is not None else None
        m = epe_metrics(est, gt, pf, fg)
        for k, v in m.items():
            if v == v:
                acc.setdefault(k, []).append(v)
    return {k: sum(v) / len(v) for k, v in acc.items()}
