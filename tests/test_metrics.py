"""Row N3 (evaluation metrics): the numpy restatement in oracle/ref_metrics.py against hand-computed answers, and the product
(deflow_amd/metrics.py, vectorised torch) against the restatement on seeded frames and on every boundary the definitions have
(the 0.05 m dynamic threshold is inclusive, the 35 m box / radius are inclusive, speed-bucket edges are left-closed)."""
import math

import numpy as np
import pytest
import torch

from oracle import ref_metrics as R


def _frame(rng, n, far=False):
    pc0 = rng.normal(0, 25 if far else 12, (n, 3)) * [1, 1, 0.1]
    rigid = rng.normal(0, 0.3, (1, 3)) + 0.01 * pc0[:, [1, 0, 2]] * [-1, 1, 0]      # a small rotation + translation field
    moving = rng.random(n) < 0.3
    gt = rigid + np.where(moving[:, None], rng.normal(0, 0.6, (n, 3)), 0.0)
    est = gt + rng.normal(0, 0.08, (n, 3)) * (rng.random((n, 1)) < 0.7)
    valid = rng.random(n) < 0.9
    cats = rng.integers(0, 31, n)
    cats[rng.random(n) < 0.4] = 0
    return est, rigid, pc0, gt, valid, cats


def test_oracle_known_answers():
    """six points, every number derived by hand"""
    pc0 = np.array([[1, 1, 0], [10, -34.9, 0], [35.0, 0, 0], [35.1, 0, 0], [-3, 4, 0], [30, 30, 0]], float)
    rigid = np.tile([0.1, 0.0, 0.0], (6, 1))
    gt = rigid + np.array([[0, 0, 0], [0.05, 0, 0], [0.0499, 0, 0], [1, 0, 0], [0, 0.5, 0], [0, 0, 0]])
    est = gt + np.array([[0.02, 0, 0], [0, 0.1, 0], [0, 0, 0.3], [5, 0, 0], [0.04, 0, 0], [0, 0.2, 0]])
    valid = np.array([1, 1, 1, 1, 1, 0], bool)
    cats = np.array([0, R.CATEGORY_TO_INDEX["REGULAR_VEHICLE"], R.CATEGORY_TO_INDEX["PEDESTRIAN"], 19, R.CATEGORY_TO_INDEX["BICYCLE"], 0])
    v1 = R.evaluate_leaderboard(est, rigid, pc0, gt, valid, cats)
    # point 0: background static close (err .02); 1: foreground DYNAMIC (|gt - rigid| = 0.05 exactly -> >=) close (err .1);
    # 2: foreground static (0.0499) close (x = 35.0 exactly -> <=) (err .3); 3: far (35.1) -> not in the table; 4: foreground dynamic
    # close (err .04); 5: invalid
    assert v1["EPE_BS"] == pytest.approx(0.02) and v1["EPE_FS"] == pytest.approx(0.3)
    assert v1["EPE_FD"] == pytest.approx((0.1 + 0.04) / 2)
    # predicted dynamic = |est - rigid| >= 0.05: p0 .02 no, p1 |(.05,.1)| yes, p2 |(.0499,0,.3)| yes, p4 |(.04,.5)| yes
    # gt dynamic: p1, p4 -> TP 2 (p1, p4), FP 1 (p2), FN 0 -> IoU 2/3
    assert v1["IoU"] == pytest.approx(2 / 3) and v1["n"] == 4
    assert v1["EPE"] == pytest.approx((0.02 + 0.1 + 0.3 + 0.04) / 4)
    # strict accuracy: err < .05 or err / |gt| < .05: p0 yes (.02), p1 no (.1 / .15), p2 no, p4 yes (.04) -> 1/2; relaxed: p1 .1 !< .1, rel .67 -> no
    assert v1["AccS"] == pytest.approx(0.5) and v1["AccR"] == pytest.approx(0.5)
    v2 = R.evaluate_leaderboard_v2(est, rigid, pc0, gt, valid, cats)
    # radius: p1 |(10, 34.9)| = 36.3 > 35 -> out; p2 35.0 in; p3 out; p5 invalid.  p0 BACKGROUND speed 0 -> bucket 0; p2 PEDESTRIAN speed
    # .0499 -> bucket 1 ([.04, .08)); p4 WHEELED_VRU speed .5 -> bucket 12 ([.48, .52))
    assert sorted(v2) == sorted([("BACKGROUND", 0, pytest.approx(0.02), pytest.approx(0.0), 1),
                                 ("PEDESTRIAN", 1, pytest.approx(0.3), pytest.approx(0.0499), 1),
                                 ("WHEELED_VRU", 12, pytest.approx(0.04), pytest.approx(0.5), 1)])
    om = R.OfficialMetrics()
    om.step(v1, v2)
    om.step(v1, v2)
    r1, r2 = om.result(1), om.result(2)
    assert r1["Three-way"] == pytest.approx((0.07 + 0.3 + 0.02) / 3) and r1["n"] == 8
    assert r2["BACKGROUND/Static"] == pytest.approx(0.02) and math.isnan(r2["BACKGROUND/Dynamic"])
    assert r2["PEDESTRIAN/Dynamic"] == pytest.approx(0.3 / 0.0499) and r2["WHEELED_VRU/Dynamic"] == pytest.approx(0.04 / 0.5)
    assert math.isnan(r2["CAR/Static"]) and r2["mean/Static"] == pytest.approx(0.02)
    assert r2["mean/Dynamic"] == pytest.approx((0.3 / 0.0499 + 0.08) / 2)
    # the space-time angle: identical flows -> 0; opposite unit flows -> angle between (1, .1) and (-1, .1)
    a = R.compute_angle_error(np.array([[1.0, 0, 0], [1.0, 0, 0]]), np.array([[1.0, 0, 0], [-1.0, 0, 0]]))
    assert a[0] == pytest.approx(0.0, abs=1e-7) and a[1] == pytest.approx(math.pi - 2 * math.atan(0.1))


def test_category_tables_agree():
    """the product's index table = the restatement's name table"""
    from deflow_amd import metrics as M
    assert len(R.ANNOTATION_CATEGORIES) == 30 and M.N_CATEGORIES == 31 and tuple(R.BUCKETED_METACATEGORIES) == M.META_CLASSES
    for ci, (name, cats) in enumerate(R.BUCKETED_METACATEGORIES.items()):
        for c in cats:
            assert M._META_OF[R.CATEGORY_TO_INDEX[c]] == ci, (name, c)
    evaluated = {R.CATEGORY_TO_INDEX[c] for cats in R.BUCKETED_METACATEGORIES.values() for c in cats}
    assert {i for i, m in enumerate(M._META_OF) if m >= 0} == evaluated
    assert M.N_BUCKETS == len(R.BUCKET_EDGES) - 1
    edges = torch.arange(1, M.N_BUCKETS, dtype=torch.float64) * M.BUCKET_WIDTH
    assert np.array_equal(edges.numpy(), R.BUCKET_EDGES[1:-1])


def _both(frames):
    from deflow_amd.metrics import OfficialMetrics
    ref, mine = R.OfficialMetrics(), OfficialMetrics()
    for est, rigid, pc0, gt, valid, cats in frames:
        ref.step(R.evaluate_leaderboard(est, rigid, pc0, gt, valid, cats), R.evaluate_leaderboard_v2(est, rigid, pc0, gt, valid, cats))
        t = lambda a, dt=torch.float64: torch.from_numpy(np.asarray(a)).to(dt)
        mine.step(t(est), t(rigid), t(pc0), t(gt), t(valid, torch.bool), t(cats, torch.long))
    return ref, mine


def _same(a, b, tol=1e-12):
    assert set(a) == set(b), (sorted(a), sorted(b))
    for k in a:
        if isinstance(a[k], float) and math.isnan(a[k]):
            assert math.isnan(b[k]), k
        else:
            # arccos is ill-conditioned at 1 (identical flows): rounding of the normalised dot product, 1e-16, becomes 1e-8 of angle
            t = max(tol, 1e-7) if k == "Angle" else tol
            assert b[k] == pytest.approx(a[k], rel=t, abs=t), k


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_product_metrics_match_the_restatement(seed):
    rng = np.random.default_rng(seed)
    frames = [_frame(rng, int(rng.integers(50, 400)), far=bool(i % 2)) for i in range(6)]
    frames[1][0][3] = np.nan                       # a non-finite estimate row, a non-finite label row
    frames[2][3][5] = np.nan
    frames.append(_frame(rng, 0))                  # an empty frame
    e = _frame(rng, 40)
    frames.append((e[0], e[1], e[2], e[3], np.zeros(40, bool), e[5]))        # nothing valid
    e = _frame(rng, 40)
    frames.append((e[0], e[1], e[2], e[3], e[4], np.zeros(40, int)))         # background only
    ref, mine = _both(frames)
    _same(ref.result(1), mine.result(1))
    _same(ref.result(2), mine.result(2))
    assert "Three-way" in mine.table(1) and "WHEELED_VRU" in mine.table(2)


def test_product_metrics_boundaries():
    """values exactly on every threshold, as exactly representable doubles"""
    n = 8
    pc0 = np.array([[35.0, 35.0, 0], [35.0, -35.0, 1], [np.nextafter(35.0, 36), 0, 0], [0, 35.0, 0], [21.0, 28.0, 0], [0, 0, 0],
                    [24.0, 26.0, 0], [1, 1, 1]], float)                     # |(21, 28)| = 35 exactly; |(24, 26)| = 35.38
    rigid = np.zeros((n, 3))
    gt = np.zeros((n, 3))
    gt[0, 0] = 0.05                                    # dynamic (>=)
    gt[1, 0] = np.nextafter(0.05, 0)                   # static
    gt[3, 1] = 0.04                                    # on the first bucket edge -> bucket 1
    gt[4, 2] = 2.0                                     # on the last edge -> the open bucket
    gt[5, 0] = np.nextafter(2.0, 0)                    # bucket 49
    gt[6, 0] = 1.0
    gt[7, 0] = 0.08
    est = gt + 0.05 * np.eye(3)[np.arange(n) % 3]      # error exactly 0.05 -> NOT strictly accurate in absolute terms
    valid = np.ones(n, bool)
    cats = np.array([19, 19, 19, 0, 17, 3, 6, 19])
    ref, mine = _both([(est, rigid, pc0, gt, valid, cats)])
    _same(ref.result(1), mine.result(1))
    _same(ref.result(2), mine.result(2))
    r2 = mine.result(2)
    assert mine.count[0, 1] == 1 and mine.count[3, 50] == 1 and mine.count[4, 49] == 1      # BACKGROUND b1, PEDESTRIAN open, VRU b49
    assert mine.count[2].sum() == 0                                                         # the BOX_TRUCK point is outside the radius
    assert r2["PEDESTRIAN/Dynamic"] == pytest.approx(0.05 / 2.0)
    r1 = mine.result(1)
    assert r1["n"] == 7                                 # all but the point a hair outside the box (the corner points are inside)


def test_evaluate_batch_feeds_the_official_tables():
    """evaluate_batch: result dict + batch (flow_is_valid, categories, eval_mask) -> OfficialMetrics on pose_flow[valid] + flow"""
    from deflow_amd.metrics import OfficialMetrics, evaluate_batch
    rng = np.random.default_rng(5)
    est, rigid, pc0, gt, valid, cats = _frame(rng, 120)
    t = lambda a, dt=torch.float32: torch.from_numpy(np.asarray(a)).to(dt)
    vi = torch.arange(0, 120, 2)                                    # the model's valid points: every second one
    res = {"flow": [t(est - rigid)[vi]], "pc0_valid_point_idxes": [vi], "pose_flow": [t(rigid)]}
    em = rng.random(120) < 0.8
    batch = {"pc0": t(pc0)[None], "flow": t(gt)[None], "flow_is_valid": t(valid, torch.bool)[None],
             "flow_category_indices": t(cats, torch.uint8)[None], "eval_mask": t(em, torch.bool)[None]}
    om = OfficialMetrics()
    m = evaluate_batch(res, batch, om)
    k = vi.numpy()
    f32 = lambda a: np.asarray(a, np.float32).astype(np.float64)
    est32 = (t(rigid)[vi] + t(est - rigid)[vi]).double().numpy()    # what the product adds up, in the product's precision
    ref = R.OfficialMetrics()
    ref.step(R.evaluate_leaderboard(est32, f32(rigid)[k], f32(pc0)[k], f32(gt)[k], (valid & em)[k], cats[k]),
             R.evaluate_leaderboard_v2(est32, f32(rigid)[k], f32(pc0)[k], f32(gt)[k], (valid & em)[k], cats[k]))
    _same(ref.result(1), om.result(1), 1e-9)
    _same(ref.result(2), om.result(2), 1e-9)
    assert m["n"] == int((valid & em)[k].sum())


def test_mixed_eval_mask_batch_scores_only_the_benchmark_frames(tmp_path):
    """upstream writes ``eval_mask`` on the official evaluation frames only, so a validation batch mixes frames with and
    without one.  collate_fn_pad handles the mask per SAMPLE (no KeyError whichever frame comes first, no silently ignored masks);
    evaluate_batch leaves the frames without a mask out of the tables; HDF5Dataset(eval=True) prefers index_eval.pkl."""
    import pickle
    from deflow_amd.data import HDF5Dataset, collate_fn_pad
    from deflow_amd.metrics import OfficialMetrics, evaluate_batch
    rng = np.random.default_rng(11)
    t = lambda a, dt=torch.float32: torch.from_numpy(np.asarray(a)).to(dt)

    def item(n, with_mask):
        est, rigid, pc0, gt, valid, cats = _frame(rng, n)
        it = {"scene_id": "s", "timestamp": 0, "pc0": t(pc0), "gm0": torch.zeros(n, dtype=torch.bool), "pose0": torch.eye(4),
              "pc1": t(pc0), "gm1": torch.zeros(n, dtype=torch.bool), "pose1": torch.eye(4), "flow": t(gt),
              "flow_is_valid": t(valid, torch.bool), "flow_category_indices": t(cats, torch.uint8)}
        it["gm0"][::7] = True                                      # a few ground points: the mask is cut like every per-point field
        if with_mask:
            it["eval_mask"] = t(rng.random(n) < 0.7, torch.bool)
        return it, est, rigid

    items = [item(90, False), item(120, True), item(60, False)]
    for order in ([0, 1, 2], [1, 0, 2]):                            # the frame WITHOUT a mask first, and the frame with one first
        batch = collate_fn_pad([items[i][0] for i in order])
        assert batch["has_eval_mask"].tolist() == [items[i][0].get("eval_mask") is not None for i in order]
        assert batch["eval_mask"].shape == batch["flow_is_valid"].shape
        for row, i in enumerate(order):
            it = items[i][0]
            keep = ~it["gm0"]
            want = it["eval_mask"][keep] if "eval_mask" in it else torch.ones(int(keep.sum()), dtype=torch.bool)
            assert torch.equal(batch["eval_mask"][row, : int(keep.sum())], want)
            assert not batch["eval_mask"][row, int(keep.sum()):].any()
        # a "model result" that keeps every non-padding point
        res = {"flow": [], "pc0_valid_point_idxes": [], "pose_flow": []}
        for row, i in enumerate(order):
            it, est, rigid = items[i]
            keep = ~it["gm0"]
            n = int(keep.sum())
            res["pc0_valid_point_idxes"].append(torch.arange(n))
            pf = torch.zeros(batch["pc0"].shape[1], 3)
            pf[:n] = t(rigid)[keep]
            res["pose_flow"].append(pf)
            res["flow"].append((t(est) - t(rigid))[keep])
        om_all, om_one = OfficialMetrics(), OfficialMetrics()
        evaluate_batch(res, batch, om_all)
        row1 = order.index(1)
        one = {k: (v[row1:row1 + 1] if isinstance(v, torch.Tensor) else [v[row1]]) for k, v in batch.items()}
        res1 = {k: [v[row1]] for k, v in res.items()}
        evaluate_batch(res1, one, om_one)
        _same(om_one.result(1), om_all.result(1), 0.0)              # only the frame with the mask was scored
        _same(om_one.result(2), om_all.result(2), 0.0)
    # a batch in which NO frame has a mask scores every frame (a split without the benchmark's masks)
    b2 = collate_fn_pad([items[0][0], items[2][0]])
    assert "eval_mask" not in b2 and "has_eval_mask" not in b2
    # index_eval.pkl wins under eval=True
    pickle.dump([["s", 1], ["s", 2], ["s", 3]], open(tmp_path / "index_total.pkl", "wb"))
    pickle.dump([["s", 2]], open(tmp_path / "index_eval.pkl", "wb"))
    assert len(HDF5Dataset(str(tmp_path))) == 3 and len(HDF5Dataset(str(tmp_path), eval=True)) == 1
    assert HDF5Dataset(str(tmp_path), eval=True).index_file == "index_eval.pkl"
