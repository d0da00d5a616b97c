"""CPU: hydra-style override parsing, Lightning-layout checkpoint round trip, EPE metrics (rows N1 / N3)."""
import os

import pytest
import torch


def test_parse_overrides_reference_commands():
    from deflow_amd.train import parse_overrides, grid_from
    c = parse_overrides("model=deflow lr=2e-4 epochs=15 batch_size=16 loss_fn=deflowLoss".split())   # [REF README.md:66]
    assert c["lr"] == 2e-4 and c["epochs"] == 15 and c["batch_size"] == 16 and grid_from(c) == [512, 512]
    c = parse_overrides(["model=deflow", "model.target.num_iters=8", "model.target.decoder_option=gru",
                         "voxel_size=[0.1, 0.1, 6]", "wandb_mode=online", "slurm_id=123"])            # [REF 1_train.sh:42,66,74]
    assert c["model.target.num_iters"] == 8 and c["voxel_size"] == [0.1, 0.1, 6] and grid_from(c) == [1024, 1024]
    assert parse_overrides(["model=fastflow3d"])["model.target.decoder_option"] == "linear"
    c = parse_overrides("model=fastflow3d lr=4e-5 epochs=20 batch_size=16 loss_fn=ff3dLoss".split())   # [REF README.md:68]
    assert c["loss_fn"] == "ff3dLoss" and c["model.target.decoder_option"] == "linear"
    assert parse_overrides(["loss_fn=zeroflowLoss"])["loss_fn"] == "zeroflowLoss"                       # [REF 1_train.sh:70]
    with pytest.raises(SystemExit):
        parse_overrides(["loss_fn=chamferLoss"])
    # the 16-iteration ablation and the 0.4 m fastflow3d resolution [REF 1_train.sh:50,78]
    assert parse_overrides(["model=deflow", "model.target.num_iters=16"])["model.target.num_iters"] == 16
    c = parse_overrides(["model=fastflow3d", "voxel_size=[0.4, 0.4, 6]"])
    assert grid_from(c) == [256, 256]


def test_dataset_path_and_strict_keys():
    """[REF 2_eval.sh:33-35; 1_train.sh:30]: dataset_path=<root> names <root>/train and <root>/val; keys that select the model
    or the optimizer are never swallowed silently (`model.target.num_iter=8` trained with the default)"""
    from deflow_amd.train import parse_overrides
    c = parse_overrides(["wandb_mode=online", "dataset_path=/scratch/local/av2/sensor", "av2_mode=val", "checkpoint=/x/y.ckpt"])
    assert c["val_data"] == "/scratch/local/av2/sensor/val" and c["train_data"] == "/scratch/local/av2/sensor/train"
    c = parse_overrides(["dataset_path=/d", "val_data=/elsewhere/val"])
    assert c["val_data"] == "/elsewhere/val" and c["train_data"] == "/d/train"
    c = parse_overrides("slurm_id=1 wandb_mode=online train_data=/s/train val_data=/s/val num_workers=16 model=deflow lr=2e-6 "
                        "epochs=50 batch_size=10 loss_fn=deflowLoss".split())                                # [REF 1_train.sh:28-30]
    assert c["train_data"] == "/s/train" and c["num_workers"] == 16 and c["batch_size"] == 10
    for bad in ("model.target.num_iter=8", "optimizer.learning_rate=1e-3", "model.target.decoder=gru", "optimizer.name=SGD"):
        with pytest.raises(SystemExit):
            parse_overrides([bad])
    assert parse_overrides(["optimizer.lr=1e-3"])["lr"] == 1e-3 and parse_overrides(["optimizer.name=Adam"])["lr"] == 2e-4
    assert parse_overrides(["model.name=fastflow3d"])["model.target.decoder_option"] == "linear"
    assert parse_overrides(["model.target.grid_feature_size=[512, 512]"])["voxel_size"] == [0.2, 0.2, 6]
    with pytest.raises(SystemExit):
        parse_overrides(["model.target.grid_feature_size=[256, 256]"])
    with pytest.raises(SystemExit):
        parse_overrides(["av2_mode=train"])


def test_checkpoint_roundtrip_lightning_layout(tmp_path):
    import deflow_amd
    from deflow_amd.optim import Trainer
    from deflow_amd.train import save_checkpoint, parse_overrides
    cfg = parse_overrides(["voxel_size=[0.2, 0.2, 6]", "point_cloud_range=[-6.4, -6.4, -3, 6.4, 6.4, 3]"])
    kw = dict(voxel_size=cfg["voxel_size"], point_cloud_range=cfg["point_cloud_range"], grid_feature_size=[64, 64])
    torch.manual_seed(1)
    m = deflow_amd.DeFlow(**kw)
    tr = Trainer(m)
    p = os.path.join(tmp_path, "x.ckpt")
    save_checkpoint(p, m, tr, cfg, 3, 42)
    ck = torch.load(p, map_location="cpu")
    assert all(k.startswith("model.") for k in ck["state_dict"]) and ck["epoch"] == 3
    assert ck["state_dict"]["model.backbone.decoder_step4.weight"].is_contiguous()
    torch.manual_seed(2)
    m2 = deflow_amd.DeFlow(**kw)
    r = m2.load_from_checkpoint(p)            # the reference's loader contract [REF deflow.py:41-47]
    assert not r.missing_keys and not r.unexpected_keys
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k


def test_epe_metrics():
    from deflow_amd.metrics import epe_metrics
    gt = torch.tensor([[0.0, 0, 0], [1.0, 0, 0], [0.0, 2.0, 0], [float("nan")] * 3])
    est = torch.tensor([[0.03, 0, 0], [1.0, 0.2, 0], [0.0, 2.0, 0.04], [0.0, 0, 0]])
    pose = torch.zeros(4, 3)
    m = epe_metrics(est, gt, pose)
    assert m["n"] == 3 and abs(m["EPE"] - (0.03 + 0.2 + 0.04) / 3) < 1e-6
    assert abs(m["AccS"] - 2 / 3) < 1e-6 and abs(m["AccR"] - 2 / 3) < 1e-6
    assert abs(m["EPE_FD"] - 0.12) < 1e-6 and abs(m["EPE_FS"] - 0.03) < 1e-6


def test_epe_metrics_use_scene_file_labels():
    """labelled batches: points whose label is invalid are left out; category 0 is background, the rest foreground"""
    import torch
    from deflow_amd.metrics import evaluate_batch
    N = 50
    res = {"flow": [torch.zeros(40, 3)], "pc0_valid_point_idxes": [torch.arange(40)], "pose_flow": [torch.zeros(N, 3)]}
    gt = torch.zeros(1, N, 3)
    gt[0, :10, 0] = 1.0                                   # ten moving points (1 m / frame), all foreground
    batch = {"flow": gt, "flow_is_valid": torch.ones(1, N, dtype=torch.bool),
             "flow_category_indices": torch.cat([torch.ones(20), torch.zeros(30)]).to(torch.uint8)[None]}
    batch["flow_is_valid"][0, 0] = False
    m = evaluate_batch(res, batch)
    assert m["n"] == 39 and m["EPE_FD"] == 1.0 and m["EPE_FS"] == 0.0 and m["EPE_BS"] == 0.0
    assert abs(m["EPE_3way"] - 1 / 3) < 1e-9 and abs(m["EPE"] - 9 / 39) < 1e-6


def test_amax_slot_pool_and_version_guard():
    """host side of the fp16x2 bound bookkeeping (no kernels): slots are distinct zero-filled views of a pool that is renewed when
    it runs out or on amax_pool_reset(); a bound left on a tensor is handed to descriptors only while nobody wrote the tensor"""
    import torch
    from deflow_amd import ops
    from deflow_amd._lib import img
    ops.amax_pool_reset()
    a, b = ops.amax_slot("cpu"), ops.amax_slot("cpu")
    assert a.shape == (1,) and float(a) == 0.0 and a.data_ptr() != b.data_ptr() and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()
    a.fill_(5.0)
    ops.amax_pool_reset()
    c = ops.amax_slot("cpu")
    assert float(c) == 0.0 and c.untyped_storage().data_ptr() != a.untyped_storage().data_ptr() and float(a) == 5.0
    for _ in range(ops._AMAX_POOL_N + 3):          # running out of slots renews the pool, old slots stay valid
        d = ops.amax_slot("cpu")
    assert float(d) == 0.0 and float(c) == 0.0
    t = torch.zeros(1, 2, 4, 8)
    t._df_amax = (a, t._version)
    assert img(t)._amax is a and not hasattr(img(t, 4, 4), "_src") and img(t)._src is t
    t.add_(1.0)                                     # an in-place torch write: the bound no longer describes the tensor
    assert not hasattr(img(t), "_amax")
