"""Minimal trainer with the reference's command-line surface:

    python -m deflow_amd.train model=deflow lr=2e-4 epochs=15 batch_size=16 loss_fn=deflowLoss \
           "model.target.num_iters=4" "voxel_size=[0.2, 0.2, 6]" checkpoint=out.ckpt      [REF README.md:66; 1_train.sh:28-78]

hydra-style ``key=value`` overrides (lists in brackets, dotted ``model.target.*`` keys), data-parallel under
``torch.distributed.run`` (one process per GPU, RCCL), checkpoints in the Lightning layout the reference's
``load_from_checkpoint`` expects ({"state_dict": {"model.<name>": tensor}, "hyper_parameters": cfg, ...}
[REF deflow.py:41-47]).  ``train_data=<dir>`` / ``val_data=<dir>`` read the preprocessed scene files the reference trains
on (``<scene>.h5`` + ``index_total.pkl``, section 8(f) N2) through deflow_amd/data.py (in-tree HDF5 reader, NaN-pad
collate, per-rank sharding, ``num_workers`` reader threads prefetching to the GPU; ``stage_dir=<scratch>`` first copies
the files node-local as 1_train.sh does); ``train_data=synthetic`` draws seeded Argoverse-2-shaped pairs
(deflow_amd/synth.py).  wandb / slurm keys are accepted and ignored."""
from __future__ import annotations

import ast
import json
import os
import sys
import time
from typing import Any, Dict, List

import torch

DEFAULTS: Dict[str, Any] = {
    "model": "deflow", "lr": 2e-4, "epochs": 1, "batch_size": 16, "loss_fn": "deflowLoss", "num_workers": 0,
    "voxel_size": [0.2, 0.2, 6], "point_cloud_range": [-51.2, -51.2, -3, 51.2, 51.2, 3],
    "model.target.num_iters": 4, "model.target.decoder_option": "gru", "gradient_clip_val": 0.0, "sync_bn": False, "dtype": "fp32", "graph": False, "dist_backend": "nccl", "resume": False,
    "train_data": "synthetic", "val_data": "synthetic", "pairs_per_epoch": 64, "points_per_cloud": 80000,
    "stage_dir": "", "checkpoint": "", "save_checkpoint": "", "seed": 20240116, "wandb_mode": "disabled", "slurm_id": "", "log_every": 50,   # Lightning's log_every_n_steps default; each log line syncs
}


# hydra / wandb / slurm plumbing of the reference's command lines [REF 1_train.sh:28-78]: no-op here, accepted silently.
# Nothing that selects data, the model or the optimizer is in this list: a mistyped ``model.target.num_iter=8`` or
# ``optimizer.lr=`` must not train with the defaults.
_IGNORED_PREFIXES = ("wandb", "slurm", "hydra", "model.val_monitor", "exp_note", "save_top_model", "val_every",
                     "leaderboard_version", "save_res", "output")
# the keys of the reference's ``model.target`` block that arrive as top-level interpolations upstream (${voxel_size} ...)
_TARGET_ALIASES = {"model.target.voxel_size": "voxel_size", "model.target.point_cloud_range": "point_cloud_range",
                   "model.name": "model", "optimizer.lr": "lr"}
DATA_KEYS = ("dataset_path", "train_data", "val_data")


def parse_overrides(argv: List[str]) -> Dict[str, Any]:
    cfg = dict(DEFAULTS)
    given = set()
    for a in argv:
        if "=" not in a:
            raise SystemExit(f"expected key=value, got {a!r}")
        k, v = a.split("=", 1)
        k = k.lstrip("+")
        try:
            val = ast.literal_eval(v)
        except (ValueError, SyntaxError):
            val = v
        k = _TARGET_ALIASES.get(k, k)
        if k == "dataset_path":
            # the reference's data root [REF 2_eval.sh:33-35; 1_train.sh:12-14]: <root>/train and <root>/val hold the scene files
            # (upstream's config interpolates train_data / val_data from it); explicit train_data= / val_data= win
            cfg["dataset_path"] = str(val)
        elif k == "av2_mode":
            if val not in ("val", "test"):
                raise SystemExit(f"unknown av2_mode {val!r} (val, test)")
            cfg["av2_mode"] = val
        elif k == "gpus":
            print(f"[deflow_amd.train] note: gpus={val} is set by the launcher here (python -m torch.distributed.run "
                  "--nproc-per-node N -m deflow_amd.train ...): one process per GPU", file=sys.stderr)
            continue
        elif k == "model.target.grid_feature_size":
            cfg["_grid_feature_size"] = list(val)        # checked against voxel_size / point_cloud_range below
            continue
        elif k == "optimizer.name":
            if str(val).lower() != "adam":
                raise SystemExit(f"optimizer.name={val!r}: this trainer implements Adam (the reference's optimizer) only")
            continue
        elif k.startswith(("model.target.", "optimizer.")) and k not in DEFAULTS:
            known = sorted(x for x in list(DEFAULTS) + list(_TARGET_ALIASES) if x.startswith(("model.target.", "optimizer.")))
            raise SystemExit(f"unknown override {k!r}; known: {', '.join(known)}, model.target.grid_feature_size")
        elif k not in DEFAULTS and not k.startswith(_IGNORED_PREFIXES):
            print(f"[deflow_amd.train] warning: unknown override {k!r} (accepted, unused); known keys: "
                  f"{', '.join(sorted(DEFAULTS))}", file=sys.stderr)
        given.add(k)
        if k not in ("dataset_path", "av2_mode"):
            cfg[k] = val
    root = cfg.get("dataset_path")
    if root:
        if "train_data" not in given:
            cfg["train_data"] = os.path.join(root, "train")
        if "val_data" not in given:
            cfg["val_data"] = os.path.join(root, "val")
    cfg["_given"] = sorted(given)
    if cfg["model"] not in ("deflow", "fastflow3d"):
        raise SystemExit(f"unknown model {cfg['model']!r}")
    if cfg["model"] == "fastflow3d":
        cfg["model.target.decoder_option"] = "linear"
    if cfg["loss_fn"] not in ("deflowLoss", "ff3dLoss", "zeroflowLoss"):
        raise SystemExit(f"unknown loss_fn {cfg['loss_fn']!r} (deflowLoss, ff3dLoss, zeroflowLoss)")
    gfs = cfg.pop("_grid_feature_size", None)
    if gfs is not None and list(gfs) != grid_from(cfg):
        raise SystemExit(f"model.target.grid_feature_size={gfs} does not match voxel_size / point_cloud_range ({grid_from(cfg)})")
    return cfg


def grid_from(cfg) -> List[int]:
    vs, rg = cfg["voxel_size"], cfg["point_cloud_range"]
    return [int(round((rg[4] - rg[1]) / vs[1])), int(round((rg[3] - rg[0]) / vs[0]))]


def build_model(cfg):
    import deflow_amd
    return deflow_amd.DeFlow(voxel_size=cfg["voxel_size"], point_cloud_range=cfg["point_cloud_range"],
                             grid_feature_size=grid_from(cfg), decoder_option=cfg["model.target.decoder_option"],
                             num_iters=int(cfg["model.target.num_iters"]))


def save_checkpoint(path: str, model, trainer, cfg, epoch: int, step: int):
    sd = {"model." + k: v.detach().cpu().contiguous() for k, v in model.state_dict().items()}
    opt = {k: (v.detach().cpu() if isinstance(v, torch.Tensor) else v) for k, v in trainer.opt.state_dict().items()}
    cfg = {k: v for k, v in cfg.items() if not k.startswith("_")}
    torch.save({"state_dict": sd, "hyper_parameters": {"cfg": cfg}, "epoch": epoch, "global_step": step,
                "optimizer_states": [opt], "pytorch-lightning_version": "deflow_amd"}, path)


def main(argv=None):
    cfg = parse_overrides(sys.argv[1:] if argv is None else argv)
    rank = int(os.environ.get("RANK", "0")); local = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available(), "training runs on the HIP engine only"
    local = local % torch.cuda.device_count()      # more ranks than GPUs only in tests (dist_backend=gloo, ranks share a device)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if cfg["dist_backend"] == "nccl":          # RCCL
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(str(cfg["dist_backend"]), rank=rank, world_size=world)
    from deflow_amd.metrics import evaluate_batch
    from deflow_amd.optim import Trainer
    from deflow_amd.synth import synth_batch

    torch.manual_seed(int(cfg["seed"]))
    model = build_model(cfg).to(dev)
    if cfg["checkpoint"]:
        model.load_from_checkpoint(cfg["checkpoint"])
    model.train()
    trainer = Trainer(model, lr=float(cfg["lr"]), loss_fn=str(cfg["loss_fn"]), gradient_clip_val=float(cfg["gradient_clip_val"]),
                      sync_bn=str(cfg["sync_bn"]).lower() in ("1", "true"), dtype=str(cfg["dtype"]))
    start_epoch, gstep0 = 0, 0
    if cfg["checkpoint"] and str(cfg["resume"]).lower() in ("1", "true"):
        # "checkpoints also include parameters and status of that epoch" [REF README.md:76-77]: continue where it stopped --
        # Adam moments and step count, epoch and global step (weights were loaded above; every rank reads the same file)
        from deflow_amd.ckpt import load_checkpoint
        ck = load_checkpoint(cfg["checkpoint"])
        if ck.get("optimizer_states"):
            trainer.opt.load_state_dict(ck["optimizer_states"][0])
            trainer.opt.lr = float(cfg["lr"])
        start_epoch, gstep0 = int(ck.get("epoch", -1)) + 1, int(ck.get("global_step", 0))
    B, N, H = int(cfg["batch_size"]), int(cfg["points_per_cloud"]), grid_from(cfg)[0]

    def scene_loader(path, shuffle):
        from deflow_amd.data import HDF5Dataset, SceneLoader, ShardedSampler, stage_to_local
        if cfg["stage_dir"]:
            dst = os.path.join(str(cfg["stage_dir"]), os.path.basename(os.path.normpath(path)))
            if local == 0:
                stage_to_local(path, dst, workers=max(4, int(cfg["num_workers"])))
            if world > 1:
                dist.barrier()
            path = dst
        ds = HDF5Dataset(path)
        sampler = ShardedSampler(len(ds), rank, world, shuffle=shuffle, seed=int(cfg["seed"]))
        return SceneLoader(ds, B, sampler, device=dev, num_workers=max(0, int(cfg["num_workers"])), drop_last=shuffle), sampler

    def synthetic_epoch(epoch):
        for it in range(steps_per_epoch):
            seed = Trainer.shard_seed(int(cfg["seed"]) + (epoch * steps_per_epoch + it) * B * world, rank, B)
            yield synth_batch(B, N, seed=seed, grid_hw=(H, H), device=dev)

    steps_per_epoch = max(1, int(cfg["pairs_per_epoch"]) // (B * world))
    train_loader = val_loader = None
    if cfg["train_data"] != "synthetic":
        train_loader, train_sampler = scene_loader(str(cfg["train_data"]), shuffle=True)
    if cfg["val_data"] != "synthetic":
        val_loader, _ = scene_loader(str(cfg["val_data"]), shuffle=False)
    use_graph = str(cfg["graph"]).lower() in ("1", "true")
    if use_graph and str(cfg["sync_bn"]).lower() in ("1", "true") and world > 1:
        use_graph = False     # SyncBatchNorm's collectives sit inside the forward and block the host
        if rank == 0:
            print("[deflow_amd.train] note: graph=true is ignored with sync_bn=true (running eager steps)", file=sys.stderr)
    gstep, log_step, log_t = gstep0, gstep0, time.perf_counter()
    for epoch in range(start_epoch, int(cfg["epochs"])):
        if train_loader is not None:
            train_sampler.set_epoch(epoch)
        for batch in (train_loader if train_loader is not None else synthetic_epoch(epoch)):
            if use_graph:
                # graph=true: the step is captured once as HIP graph(s) on the first batch (capture restores parameters, Adam
                # state and BatchNorm buffers after its warm-up launches: the replay below IS the first training step) and
                # replayed for every batch of the same shapes; other shapes run eagerly.  Data-parallel ranks replay graph
                # segments split at the gradient buckets (optim.SegmentedCapture)
                if getattr(trainer, "_graph", None) is None:
                    trainer.capture(batch)
                    loss = trainer.step_captured()      # (capture records the launches, it does not run them)
                else:
                    try:
                        loss = trainer.step_captured(batch)
                    except ValueError:
                        loss = trainer.step(batch)
            else:
                loss = trainer.step(batch)
            gstep += 1
            if rank == 0 and gstep % int(cfg["log_every"]) == 0:
                lv = float(loss)  # reads the loss back: the only host sync of the loop, so the rate below is a true one
                now = time.perf_counter()
                print(json.dumps({"epoch": epoch, "step": gstep, "trainer/loss": lv / B,
                                  "pairs_per_s": B * world * (gstep - log_step) / (now - log_t)}), flush=True)
                log_t, log_step = now, gstep
        trainer.sync_buffers()         # every rank validates (and rank 0 saves) rank 0's BatchNorm statistics, as DDP does
        model.eval()
        with torch.no_grad():
            if val_loader is not None:
                per_batch = [evaluate_batch(model(vb), vb) for vb in val_loader]
                metrics = {k: float(sum(m[k] for m in per_batch) / max(len(per_batch), 1)) for k in (per_batch[0] if per_batch else {})}
            else:
                vb = synth_batch(min(B, 4), N, seed=int(cfg["seed"]) + 10 ** 6 + epoch, grid_hw=(H, H), device=dev)
                metrics = evaluate_batch(model(vb), vb)
        model.train()
        if world > 1:                  # mean over the ranks that have the metric (a fixed key list: every rank reduces alike)
            keys = ["EPE", "AccS", "AccR", "n", "EPE_FD", "EPE_FS", "EPE_BS", "EPE_3way"]
            vals = [float(metrics.get(k, float("nan"))) for k in keys]
            t = torch.tensor([0.0 if v != v else v for v in vals] + [0.0 if v != v else 1.0 for v in vals], dtype=torch.float64, device=dev)
            dist.all_reduce(t)
            metrics = {k: float(t[i] / t[len(keys) + i]) for i, k in enumerate(keys) if float(t[len(keys) + i]) > 0}
        if rank == 0:
            print(json.dumps({"epoch": epoch, "val": metrics}), flush=True)
            if cfg["save_checkpoint"]:
                save_checkpoint(cfg["save_checkpoint"], model, trainer, cfg, epoch, gstep)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
