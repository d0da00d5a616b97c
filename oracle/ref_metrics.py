"""CPU restatement of the evaluation metrics the reference prints from ``eval.py ... av2_mode=val`` [REF README.md:88-91;
assets/slurm/2_eval.sh:33-35] -- TEST INFRASTRUCTURE ONLY (tests/ and nothing else import this file).

PARITY UNPINNED: the reference's metric code lives in the OpenSceneFlow submodule (``src/utils/eval_metric.py`` +
``src/utils/av2_eval.py``, which follow av2-api ``av2/evaluation/scene_flow/eval.py`` and the bucketed evaluation of Khatri et
al., "I Can't Believe It's Not Scene Flow!") and is absent from /root/reference [REF .gitmodules:1-3].  What is restated here
is the published algorithm, numpy and plain loops, one function per upstream function:

  * compute_end_point_error / compute_accuracy(strict 0.05, relax 0.10; absolute OR relative, relative = err / (|gt| + 1e-10))
    / compute_angle_error (4-D space-time vectors with a 0.1 time component)                     -- av2 eval.py
  * compute_metrics: class (Background = category 0, Foreground = every annotated category) x motion (Dynamic / Static, the
    caller's is_dynamic) x distance (Close = both |x|, |y| <= 35 m, Far), masked by is_valid; mean EPE / accuracies / angle
    error and the point count per cell                                                           -- av2 eval.py
  * evaluate_leaderboard (leaderboard_version = 1): is_dynamic = |gt - rigid| >= 0.05 m per frame (0.5 m/s at 10 Hz), predicted
    dynamic likewise; EPE_FD / EPE_FS / EPE_BS = the Close cells' mean EPE, Three-way = their mean, IoU of the dynamic
    segmentation over the Close valid points                                                     -- OpenSceneFlow av2_eval.py
  * evaluate_leaderboard_v2 (leaderboard_version = 2): points within 35 m of the sensor in xy (Euclidean), valid and finite;
    ego motion removed from both flows; per meta-class (BACKGROUND, CAR, OTHER_VEHICLES, PEDESTRIAN, WHEELED_VRU) and speed
    bucket (51 edges 0 .. 2.0 m per frame + an open last bucket) the mean EPE, mean speed and count
  * OfficialMetrics: version 1 averages the per-frame values (NaN cells skipped); version 2 accumulates count-weighted
    bucket means over frames, normalises every dynamic bucket's EPE by its mean speed, and reports per class the static EPE
    (first bucket) and the mean normalised EPE of the non-empty dynamic buckets, then their means over classes.
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import numpy as np

CLOSE_DISTANCE_THRESHOLD = 35.0
EPS = 1e-10

# av2 annotation categories, alphabetical (av2 ``AnnotationCategories``); label index = position + 1, 0 = no annotation ("NONE")
ANNOTATION_CATEGORIES = [
    "ANIMAL", "ARTICULATED_BUS", "BICYCLE", "BICYCLIST", "BOLLARD", "BOX_TRUCK", "BUS", "CONSTRUCTION_BARREL", "CONSTRUCTION_CONE",
    "DOG", "LARGE_VEHICLE", "MESSAGE_BOARD_TRAILER", "MOBILE_PEDESTRIAN_CROSSING_SIGN", "MOTORCYCLE", "MOTORCYCLIST",
    "OFFICIAL_SIGNALER", "PEDESTRIAN", "RAILED_VEHICLE", "REGULAR_VEHICLE", "SCHOOL_BUS", "SIGN", "STOP_SIGN", "STROLLER",
    "TRAFFIC_LIGHT_TRAILER", "TRUCK", "TRUCK_CAB", "VEHICULAR_TRAILER", "WHEELCHAIR", "WHEELED_DEVICE", "WHEELED_RIDER"]
CATEGORY_TO_INDEX = {"NONE": 0, **{k: i + 1 for i, k in enumerate(ANNOTATION_CATEGORIES)}}
FOREGROUND_BACKGROUND = {"Background": [0], "Foreground": list(range(1, len(ANNOTATION_CATEGORIES) + 1))}
BUCKETED_METACATEGORIES = {
    "BACKGROUND": ["NONE"],
    "CAR": ["REGULAR_VEHICLE"],
    "OTHER_VEHICLES": ["BOX_TRUCK", "LARGE_VEHICLE", "RAILED_VEHICLE", "TRUCK", "TRUCK_CAB", "VEHICULAR_TRAILER", "ARTICULATED_BUS",
                       "BUS", "SCHOOL_BUS"],
    "PEDESTRIAN": ["PEDESTRIAN", "STROLLER", "WHEELCHAIR", "OFFICIAL_SIGNALER"],
    "WHEELED_VRU": ["BICYCLE", "BICYCLIST", "MOTORCYCLE", "MOTORCYCLIST", "WHEELED_DEVICE", "WHEELED_RIDER"],
}
BUCKET_EDGES = np.concatenate([np.linspace(0.0, 2.0, 51), [np.inf]])     # 51 buckets: [0, .04), ..., [1.96, 2.0), [2.0, inf)


def compute_end_point_error(dts: np.ndarray, gts: np.ndarray) -> np.ndarray:
    return np.linalg.norm(dts - gts, axis=-1)


def compute_accuracy(dts: np.ndarray, gts: np.ndarray, threshold: float) -> np.ndarray:
    l2 = np.linalg.norm(dts - gts, axis=-1)
    rel = l2 / (np.linalg.norm(gts, axis=-1) + EPS)
    return np.logical_or(l2 < threshold, rel < threshold).astype(np.float64)


def compute_angle_error(dts: np.ndarray, gts: np.ndarray) -> np.ndarray:
    d4 = np.pad(dts, ((0, 0), (0, 1)), constant_values=0.1)
    g4 = np.pad(gts, ((0, 0), (0, 1)), constant_values=0.1)
    d4 = d4 / np.linalg.norm(d4, axis=-1, keepdims=True)
    g4 = g4 / np.linalg.norm(g4, axis=-1, keepdims=True)
    return np.arccos(np.clip((d4 * g4).sum(-1), -1.0, 1.0))


def compute_metrics(pred_flow, pred_dynamic, gts, category_indices, is_dynamic, is_close, is_valid) -> Dict[str, list]:
    """one row per (class, motion, distance) cell, in the order of the three nested loops"""
    pred_flow = np.asarray(pred_flow, np.float64)
    gts = np.asarray(gts, np.float64)
    out: Dict[str, list] = {k: [] for k in ("Class", "Motion", "Distance", "Count", "EPE", "ACCURACY_STRICT", "ACCURACY_RELAX",
                                            "ANGLE_ERROR", "TP", "TN", "FP", "FN")}
    for cls, ids in FOREGROUND_BACKGROUND.items():
        class_mask = np.isin(category_indices, ids)
        for motion, m_mask in (("Dynamic", is_dynamic), ("Static", ~is_dynamic)):
            for dist, d_mask in (("Close", is_close), ("Far", ~is_close)):
                mask = class_mask & m_mask & d_mask & is_valid
                cnt = int(mask.sum())
                out["Class"].append(cls); out["Motion"].append(motion); out["Distance"].append(dist); out["Count"].append(cnt)
                if cnt > 0:
                    p, g = pred_flow[mask], gts[mask]
                    out["EPE"].append(float(compute_end_point_error(p, g).mean()))
                    out["ACCURACY_STRICT"].append(float(compute_accuracy(p, g, 0.05).mean()))
                    out["ACCURACY_RELAX"].append(float(compute_accuracy(p, g, 0.10).mean()))
                    out["ANGLE_ERROR"].append(float(compute_angle_error(p, g).mean()))
                    pd, gd = pred_dynamic[mask], is_dynamic[mask]
                    out["TP"].append(int((pd & gd).sum())); out["TN"].append(int((~pd & ~gd).sum()))
                    out["FP"].append(int((pd & ~gd).sum())); out["FN"].append(int((~pd & gd).sum()))
                else:
                    for k in ("EPE", "ACCURACY_STRICT", "ACCURACY_RELAX", "ANGLE_ERROR"):
                        out[k].append(float("nan"))
                    for k in ("TP", "TN", "FP", "FN"):
                        out[k].append(0)
    return out


def _finite_rows(*arrs) -> np.ndarray:
    m = np.ones(arrs[0].shape[0], bool)
    for a in arrs:
        m &= np.isfinite(a).all(axis=1)
    return m


def evaluate_leaderboard(est_flow, rigid_flow, pc0, gt_flow, is_valid, pts_ids) -> Dict[str, float]:
    """leaderboard_version = 1: three-way EPE + dynamic IoU of one frame (total flows: ego motion included)"""
    est_flow, rigid_flow, pc0, gt_flow = (np.asarray(a, np.float64) for a in (est_flow, rigid_flow, pc0, gt_flow))
    ok = _finite_rows(est_flow, rigid_flow, pc0[:, :3], gt_flow)
    est_flow, rigid_flow, pc0, gt_flow = est_flow[ok], rigid_flow[ok], pc0[ok], gt_flow[ok]
    is_valid, pts_ids = np.asarray(is_valid, bool)[ok], np.asarray(pts_ids)[ok]
    gt_dyn = np.linalg.norm(gt_flow - rigid_flow, axis=-1) >= 0.05
    est_dyn = np.linalg.norm(est_flow - rigid_flow, axis=-1) >= 0.05
    is_close = np.all(np.abs(pc0[:, :2]) <= CLOSE_DISTANCE_THRESHOLD, axis=1)
    res = compute_metrics(est_flow, est_dyn, gt_flow, pts_ids, gt_dyn, is_close, is_valid)
    cell = {(c, m, d): i for i, (c, m, d) in enumerate(zip(res["Class"], res["Motion"], res["Distance"]))}
    fd, fs, bs = cell[("Foreground", "Dynamic", "Close")], cell[("Foreground", "Static", "Close")], cell[("Background", "Static", "Close")]
    tp = sum(res["TP"][i] for (c, m, d), i in cell.items() if d == "Close")
    fp = sum(res["FP"][i] for (c, m, d), i in cell.items() if d == "Close")
    fn = sum(res["FN"][i] for (c, m, d), i in cell.items() if d == "Close")
    out = {"EPE_FD": res["EPE"][fd], "EPE_FS": res["EPE"][fs], "EPE_BS": res["EPE"][bs],
           "IoU": tp / (tp + fp + fn) if (tp + fp + fn) > 0 else float("nan")}
    # whole-frame figures over the valid Close points (what the table's header line prints beside the three-way numbers)
    m = is_valid & is_close
    if m.any():
        out["EPE"] = float(compute_end_point_error(est_flow[m], gt_flow[m]).mean())
        out["AccS"] = float(compute_accuracy(est_flow[m], gt_flow[m], 0.05).mean())
        out["AccR"] = float(compute_accuracy(est_flow[m], gt_flow[m], 0.10).mean())
        out["Angle"] = float(compute_angle_error(est_flow[m], gt_flow[m]).mean())
    else:
        out.update(EPE=float("nan"), AccS=float("nan"), AccR=float("nan"), Angle=float("nan"))
    out["n"] = int(m.sum())
    return out


def evaluate_leaderboard_v2(est_flow, rigid_flow, pc0, gt_flow, is_valid, pts_ids) -> List[Tuple[str, int, float, float, int]]:
    """leaderboard_version = 2: (meta-class, bucket index, mean EPE, mean speed, count) of every non-empty cell of one frame"""
    est_flow, rigid_flow, pc0, gt_flow = (np.asarray(a, np.float64) for a in (est_flow, rigid_flow, pc0, gt_flow))
    ok = _finite_rows(est_flow, rigid_flow, pc0[:, :3], gt_flow) & np.asarray(is_valid, bool)
    ok &= np.linalg.norm(pc0[:, :2], axis=-1) <= CLOSE_DISTANCE_THRESHOLD
    est = est_flow[ok] - rigid_flow[ok]
    gt = gt_flow[ok] - rigid_flow[ok]
    ids = np.asarray(pts_ids)[ok]
    speed = np.linalg.norm(gt, axis=-1)
    err = np.linalg.norm(est - gt, axis=-1)
    rows = []
    for name, cats in BUCKETED_METACATEGORIES.items():
        cat_mask = np.isin(ids, [CATEGORY_TO_INDEX[c] for c in cats])
        for bi in range(len(BUCKET_EDGES) - 1):
            m = cat_mask & (speed >= BUCKET_EDGES[bi]) & (speed < BUCKET_EDGES[bi + 1])
            cnt = int(m.sum())
            if cnt:
                rows.append((name, bi, float(err[m].mean()), float(speed[m].mean()), cnt))
    return rows


class OfficialMetrics:
    """frame-by-frame accumulation and the final table"""

    def __init__(self):
        self.v1: Dict[str, List[float]] = {}
        nb = len(BUCKET_EDGES) - 1
        self.epe = {c: np.zeros(nb) for c in BUCKETED_METACATEGORIES}
        self.speed = {c: np.zeros(nb) for c in BUCKETED_METACATEGORIES}
        self.count = {c: np.zeros(nb, np.int64) for c in BUCKETED_METACATEGORIES}

    def step(self, v1: Dict[str, float], v2: List[Tuple[str, int, float, float, int]]):
        for k, v in v1.items():
            if k == "n" or not math.isnan(v):
                self.v1.setdefault(k, []).append(v)
        for name, bi, e, s, n in v2:
            c0 = self.count[name][bi]
            self.epe[name][bi] = (self.epe[name][bi] * c0 + e * n) / (c0 + n)      # count-weighted running means
            self.speed[name][bi] = (self.speed[name][bi] * c0 + s * n) / (c0 + n)
            self.count[name][bi] = c0 + n

    def result(self, leaderboard_version: int = 1) -> Dict[str, float]:
        if leaderboard_version == 1:
            out = {k: (float(np.sum(v)) if k == "n" else float(np.mean(v))) for k, v in self.v1.items()}
            for k in ("EPE_FD", "EPE_FS", "EPE_BS", "IoU", "EPE", "AccS", "AccR", "Angle"):     # a cell no frame ever filled
                out.setdefault(k, float("nan"))
            out.setdefault("n", 0.0)
            three = [out[k] for k in ("EPE_FD", "EPE_FS", "EPE_BS") if k in out]
            out["Three-way"] = float(np.mean(three)) if len(three) == 3 else float("nan")
            return out
        out = {}
        stat, dyn = [], []
        for c in BUCKETED_METACATEGORIES:
            s = self.epe[c][0] if self.count[c][0] > 0 else float("nan")
            d = [self.epe[c][b] / self.speed[c][b] for b in range(1, len(self.count[c])) if self.count[c][b] > 0]
            d = float(np.mean(d)) if d else float("nan")
            out[f"{c}/Static"], out[f"{c}/Dynamic"] = float(s), d
            stat.append(s); dyn.append(d)
        out["mean/Static"] = float(np.nanmean(stat)) if not all(math.isnan(x) for x in stat) else float("nan")
        out["mean/Dynamic"] = float(np.nanmean(dyn)) if not all(math.isnan(x) for x in dyn) else float("nan")
        return out
