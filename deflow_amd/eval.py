"""``python -m deflow_amd.eval checkpoint=<ckpt> av2_mode=val dataset_path=<root>``: the reference's evaluation entry for this
model plugin ([REF README.md:88]: "python eval.py checkpoint=... av2_mode=val  # it will directly prints all metric";
[REF assets/slurm/2_eval.sh:33-35]: ``eval.py wandb_mode=online dataset_path=/scratch/local/av2/sensor av2_mode=val checkpoint=...``
-> the scene files under ``<dataset_path>/val``; ``val_data=<dir>`` names the directory directly).

The checkpoint carries its training configuration (``hyper_parameters``, as Lightning checkpoints do), so only the
checkpoint and the data need naming.  Metrics: deflow_amd/metrics.py -- ``leaderboard_version=1`` (default) the Argoverse-2 three-way
EPE table, ``leaderboard_version=2`` the bucketed normalised EPE; plus a range-free EPE / accuracy summary -- over the sweeps
of ``val_data`` (preprocessed scene files, deflow_amd/data.py) or over seeded synthetic pairs with ``val_data=synthetic``.
``av2_mode=test`` (leaderboard submission zips) is the reference's control plane and is not built."""
from __future__ import annotations

import json
import os
import sys

import torch

from .train import DATA_KEYS, DEFAULTS, _TARGET_ALIASES, build_model, grid_from, parse_overrides


def main(argv=None):
    args = list(sys.argv[1:] if argv is None else argv)
    given = {a.split("=", 1)[0].lstrip("+"): a for a in args if "=" in a}
    if given.get("av2_mode", "av2_mode=val") != "av2_mode=val":
        raise SystemExit("only av2_mode=val is implemented (test-split submission files are out of scope)")
    if "checkpoint" not in given:
        raise SystemExit("usage: python -m deflow_amd.eval checkpoint=<path> [av2_mode=val] [val_data=<dir>|synthetic]")
    path = given["checkpoint"].split("=", 1)[1]
    from .ckpt import flatten, load_checkpoint, plain
    ckpt = load_checkpoint(path)
    # configuration: defaults < what the checkpoint was trained with < what this command line names.  The reference's
    # Lightning checkpoints hold a NESTED (omegaconf) config -- model.target.num_iters etc. -- so it is flattened to the
    # dotted keys used here; keys this trainer does not know are ignored.
    cfg = dict(DEFAULTS)
    hp = plain(ckpt.get("hyper_parameters", {}))
    saved = flatten(hp.get("cfg", hp) if isinstance(hp, dict) else {})
    # nested reference configs name the architecture ``model.name`` (and repeat voxel_size / point_cloud_range under
    # ``model.target``): map them onto this trainer's keys BEFORE filtering, or a fastflow3d checkpoint is built as DeFlow and
    # loaded with strict=False
    saved = {_TARGET_ALIASES.get(k, k): v for k, v in saved.items()}
    cfg.update({k: v for k, v in saved.items() if k in DEFAULTS and k not in DATA_KEYS and not isinstance(v, dict)})
    typed = parse_overrides([a for k, a in given.items() if k not in ("leaderboard_version", "inference_dtype")])
    cfg.update({k: typed[k] for k in typed["_given"] if k in typed})
    for k in ("train_data", "val_data"):       # dataset_path=<root> [REF 2_eval.sh:33-35] -> <root>/val
        if typed[k] != DEFAULTS[k]:
            cfg[k] = typed[k]
    if cfg["model"] == "fastflow3d":
        cfg["model.target.decoder_option"] = "linear"
    if not any(k in given for k in DATA_KEYS):
        print("[deflow_amd.eval] NOTE: no dataset_path= / val_data= given -- evaluating on seeded SYNTHETIC pairs, not on a "
              "dataset (the reference reads its config's dataset_path default here)", file=sys.stderr)
    elif cfg["val_data"] == "synthetic" and given.get("val_data") != "val_data=synthetic":
        raise SystemExit("a data key was given but no validation directory resulted from it; refusing to fall back to synthetic data")
    assert torch.cuda.is_available(), "evaluation runs on the HIP engine only"
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    from .metrics import OfficialMetrics, evaluate_batch
    version = int(given.get("leaderboard_version", "leaderboard_version=1").split("=", 1)[1])
    if version not in (1, 2):
        raise SystemExit("leaderboard_version must be 1 (three-way EPE) or 2 (bucketed normalised EPE)")
    official = OfficialMetrics()
    model = build_model(cfg).to(dev)
    res = model.load_from_checkpoint(path)
    if res.missing_keys or res.unexpected_keys:   # strict=False as the reference loads [REF deflow.py:47] -- but never silently
        print(f"[deflow_amd.eval] WARNING: checkpoint / model mismatch (model={cfg['model']}): {len(res.missing_keys)} missing keys "
              f"{res.missing_keys[:4]}..., {len(res.unexpected_keys)} unexpected keys {res.unexpected_keys[:4]}...", file=sys.stderr)
    model.eval()
    if "inference_dtype" in given:
        model.inference_dtype = given["inference_dtype"].split("=", 1)[1]
    B = int(cfg["batch_size"])
    if cfg["val_data"] != "synthetic":
        from .data import HDF5Dataset, SceneLoader, ShardedSampler
        ds = HDF5Dataset(str(cfg["val_data"]), eval=True)      # index_eval.pkl (the benchmark's frames) when the directory has one
        batches = SceneLoader(ds, B, ShardedSampler(len(ds), shuffle=False), device=dev,
                              num_workers=max(0, int(cfg["num_workers"])), drop_last=False)
    else:
        from .synth import synth_batch
        H = grid_from(cfg)[0]
        batches = (synth_batch(B, int(cfg["points_per_cloud"]), seed=int(cfg["seed"]) + 10 ** 6 + i * B, grid_hw=(H, H), device=dev)
                   for i in range(max(1, int(cfg["pairs_per_epoch"]) // B)))
    tot, wsum = {}, {}
    with torch.no_grad():
        for batch in batches:
            m = evaluate_batch(model(batch), batch, official)
            w = len(batch["pose0"])
            for k, v in m.items():
                tot[k] = tot.get(k, 0.0) + v * w
                wsum[k] = wsum.get(k, 0) + w
    out = {k: tot[k] / wsum[k] for k in tot}
    # the leaderboard table the reference prints at the end of validation [REF README.md:88-91]: leaderboard_version=1 the three-way
    # EPE (+ IoU, accuracies, angle error inside the 35 m box), leaderboard_version=2 the bucketed normalised EPE
    board = official.result(version)
    print(official.table(version), file=sys.stderr, flush=True)
    print(json.dumps({"checkpoint": path, "model": cfg["model"], "val_data": cfg["val_data"], "metrics": out,
                      "leaderboard_version": version, "leaderboard": board}), flush=True)
    out = dict(out)
    out["leaderboard"] = board
    return out


if __name__ == "__main__":
    main()
