"""Generate golden vectors by EXECUTING the real reference (this container only).

Run:  python oracle/gen_golden.py           (writes tests/golden/*.npz)

Imports ``/root/reference/decoder.py`` unmodified under a synthetic parent package (the file
does ``from . import ConvWithNorms`` and then re-defines it, [REF decoder.py:4,202]) with
bytecode writing disabled (the reference tree is read-only).  ``/root/reference/deflow.py`` is
imported with the oracle's own embedder/UNet injected for the absent ``.basic.*`` modules and a
no-op ``dztimer``; that run pins ORCHESTRATION only (ego-motion arithmetic, hand-offs, result
dict) -- it is not reference parity for the injected parts.

Fixtures hold data only: weights, inputs, expected outputs and expected gradients.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
sys.path.insert(0, os.path.dirname(HERE))


def _load_ref_decoder():
    pkg = types.ModuleType("_refpkg")
    pkg.__path__ = []  # mark as package
    pkg.ConvWithNorms = object  # placeholder; decoder.py re-defines the class itself
    sys.modules["_refpkg"] = pkg
    spec = importlib.util.spec_from_file_location("_refpkg.decoder", os.path.join(REF, "decoder.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["_refpkg.decoder"] = mod
    spec.loader.exec_module(mod)
    return mod


def _load_ref_deflow(refdec):
    from oracle import ref_torch as O

    class _Timer:  # records nothing; same call surface as dztimer.Timing [REF deflow.py:38-95]
        def start(self, *_a):
            pass

        def stop(self):
            pass

        def __getitem__(self, _i):
            return self

    dz = types.ModuleType("dztimer")
    dz.Timing = _Timer
    sys.modules["dztimer"] = dz
    top = types.ModuleType("_refmodels")
    top.__path__ = []
    basic = types.ModuleType("_refmodels.basic")
    basic.__path__ = []
    basic.cal_pose0to1 = O.cal_pose0to1
    unet = types.ModuleType("_refmodels.basic.unet")
    unet.FastFlow3DUNet = O.FastFlow3DUNet
    enc = types.ModuleType("_refmodels.basic.encoder")
    enc.DynamicEmbedder = O.DynamicEmbedder
    dec = types.ModuleType("_refmodels.basic.decoder")
    dec.LinearDecoder = refdec.LinearDecoder
    dec.ConvGRUDecoder = refdec.ConvGRUDecoder
    for name, m in (("_refmodels", top), ("_refmodels.basic", basic), ("_refmodels.basic.unet", unet),
                    ("_refmodels.basic.encoder", enc), ("_refmodels.basic.decoder", dec)):
        sys.modules[name] = m
    spec = importlib.util.spec_from_file_location("_refmodels.deflow", os.path.join(REF, "deflow.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["_refmodels.deflow"] = mod
    spec.loader.exec_module(mod)
    return mod


def _np(d):
    return {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in d.items()}


def _state(mod, prefix="w."):
    return {prefix + k: v.detach().clone() for k, v in mod.state_dict().items()}


def synth_pair(seed: int, n: int, grid_hw=(512, 512), nan_frac=0.02):
    """Small AV2-shaped pair (same recipe as deflow_amd.synth, kept local so fixtures do not depend on product code)."""
    g = torch.Generator().manual_seed(seed)
    sigma = 20.0 * grid_hw[0] / 512.0
    xy = torch.randn(n, 2, generator=g) * sigma
    z = (torch.rand(n, 1, generator=g) * 6.6) - 3.3
    pc0 = torch.cat([xy, z], 1)
    yaw = (torch.rand(1, generator=g).item() * 4 - 2) * np.pi / 180
    T = torch.eye(4)
    T[0, 0] = np.cos(yaw); T[0, 1] = -np.sin(yaw); T[1, 0] = np.sin(yaw); T[1, 1] = np.cos(yaw)
    T[0, 3] = torch.rand(1, generator=g).item() * 1.5
    dyn = torch.rand(n, generator=g) < 0.1
    flow = torch.zeros(n, 3)
    d = torch.randn(n, 3, generator=g)
    d = d / d.norm(dim=1, keepdim=True) * (torch.rand(n, 1, generator=g) * 2.0)
    flow[dyn] = d[dyn]
    pc1 = pc0 @ T[:3, :3].T + T[:3, 3] + flow + torch.randn(n, 3, generator=g) * 0.02
    k = int(n * nan_frac)
    if k:
        pc0[-k:] = float("nan")
        pc1[-k:] = float("nan")
    gt_flow = (pc0 @ T[:3, :3].T + T[:3, 3] - pc0) + flow  # total flow incl. ego motion
    return pc0, pc1, T, gt_flow


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(1)
    R = _load_ref_decoder()

    # ---- G1 ConvGRU --------------------------------------------------------------------
    torch.manual_seed(101)
    gru = R.ConvGRU(64, 128)
    h = torch.randn(257, 128, 1, requires_grad=True)
    x = torch.randn(257, 64, 1, requires_grad=True)
    out = gru(h, x)
    gout = torch.randn_like(out)
    out.backward(gout)
    d = {"h": h, "x": x, "out": out, "gout": gout, "gh": h.grad, "gx": x.grad}
    d.update(_state(gru))
    d.update({"gw." + k: p.grad for k, p in gru.named_parameters()})
    np.savez_compressed(os.path.join(OUT, "g1_convgru.npz"), **_np(d))

    # ---- G2 ConvGRUDecoder (iters 1,4,8 and 16 = the fastflow3d-ablation / 1_train.sh:50 setting; duplicate cells; N_b in {0,1,333})
    for iters in (1, 4, 8, 16):
        torch.manual_seed(200 + iters)
        dec = R.ConvGRUDecoder(num_iters=iters)
        Hh = Ww = 16
        before = torch.randn(3, 64, Hh, Ww, requires_grad=True)
        after = torch.randn(3, 64, Hh, Ww, requires_grad=True)
        infos = []
        for nb in (333, 0, 1):
            yx = torch.randint(0, Hh, (nb, 2))
            if nb >= 5:
                yx[:5] = torch.tensor([7, 9])  # five points on one cell
            vc = torch.cat([torch.zeros(nb, 1, dtype=torch.long), yx], 1).to(torch.int32)
            infos.append({"voxel_coords": vc, "point_offsets": (torch.rand(nb, 3) - 0.5) * 0.2})
        flows = dec(before, after, infos)
        gfl = [torch.randn_like(f) for f in flows]
        sum((f * g).sum() for f, g in zip(flows, gfl)).backward()
        d = {"before": before, "after": after, "gbefore": before.grad, "gafter": after.grad, "num_iters": iters}
        for i, (info, f, g) in enumerate(zip(infos, flows, gfl)):
            d[f"vc{i}"] = info["voxel_coords"]; d[f"off{i}"] = info["point_offsets"]; d[f"flow{i}"] = f; d[f"gflow{i}"] = g
        d.update(_state(dec))
        d.update({"gw." + k: p.grad for k, p in dec.named_parameters()})
        np.savez_compressed(os.path.join(OUT, f"g2_grudecoder_it{iters}.npz"), **_np(d))

    # ---- G3 LinearDecoder ----------------------------------------------------------------
    torch.manual_seed(300)
    dec = R.LinearDecoder()
    before = torch.randn(2, 64, 16, 16, requires_grad=True)
    after = torch.randn(2, 64, 16, 16, requires_grad=True)
    infos = []
    for nb in (97, 40):
        yx = torch.randint(0, 16, (nb, 2))
        infos.append({"voxel_coords": torch.cat([torch.zeros(nb, 1, dtype=torch.long), yx], 1).to(torch.int32),
                      "point_offsets": (torch.rand(nb, 3) - 0.5) * 0.2})
    flows = dec(before, after, infos)
    gfl = [torch.randn_like(f) for f in flows]
    sum((f * g).sum() for f, g in zip(flows, gfl)).backward()
    d = {"before": before, "after": after, "gbefore": before.grad, "gafter": after.grad}
    for i, (info, f, g) in enumerate(zip(infos, flows, gfl)):
        d[f"vc{i}"] = info["voxel_coords"]; d[f"off{i}"] = info["point_offsets"]; d[f"flow{i}"] = f; d[f"gflow{i}"] = g
    d.update(_state(dec))
    d.update({"gw." + k: p.grad for k, p in dec.named_parameters()})
    np.savez_compressed(os.path.join(OUT, "g3_lineardecoder.npz"), **_np(d))

    # ---- G4 ConvWithNorms: train, eval, stride 2, and the 1x1-output BN-skip case ----------
    for tag, (cin, cout, k, s, p, hw, train) in {
        "train_s1": (32, 64, 3, 1, 1, 16, True), "train_s2": (32, 64, 3, 2, 1, 16, True),
        "eval_s1": (64, 64, 3, 1, 1, 12, False), "skip1x1": (32, 64, 3, 2, 1, 2, True),
    }.items():
        torch.manual_seed(400 + len(tag))
        m = R.ConvWithNorms(cin, cout, k, s, p)
        with torch.no_grad():
            m.batchnorm.weight.uniform_(0.5, 1.5); m.batchnorm.bias.uniform_(-0.3, 0.3)
            m.batchnorm.running_mean.uniform_(-0.2, 0.2); m.batchnorm.running_var.uniform_(0.5, 1.5)
        m.train(train)
        st0 = _state(m, "w0.")
        x = torch.randn(3, cin, hw, hw, requires_grad=True)
        y = m(x)
        gy = torch.randn_like(y)
        y.backward(gy)
        d = {"x": x, "y": y, "gy": gy, "gx": x.grad, "train": train, "k": k, "s": s, "p": p}
        d.update(st0); d.update(_state(m, "w1."))  # w1.* = state after the call (running stats)
        d.update({"gw." + kk: pp.grad for kk, pp in m.named_parameters() if pp.grad is not None})
        np.savez_compressed(os.path.join(OUT, f"g4_convwithnorms_{tag}.npz"), **_np(d))

    # ---- G5 DeFlow.forward orchestration (reference deflow.py driving injected blocks) -----
    D = _load_ref_deflow(R)
    from oracle import ref_torch as O
    torch.manual_seed(500)
    model = D.DeFlow(voxel_size=[0.2, 0.2, 6], point_cloud_range=[-6.4, -6.4, -3, 6.4, 6.4, 3],
                     grid_feature_size=[64, 64], decoder_option="gru", num_iters=2)
    model.eval()
    pcs = [synth_pair(900 + b, 600, grid_hw=(64, 64)) for b in range(2)]
    batch = {"pc0": torch.stack([p[0] for p in pcs]), "pc1": torch.stack([p[1] for p in pcs]),
             "pose0": torch.stack([torch.eye(4) for _ in pcs]),
             "pose1": torch.stack([torch.linalg.inv(p[2]) for p in pcs])}
    with torch.no_grad():
        res = model(batch)
        mine = O.DeFlow(voxel_size=[0.2, 0.2, 6], point_cloud_range=[-6.4, -6.4, -3, 6.4, 6.4, 3],
                        grid_feature_size=[64, 64], decoder_option="gru", num_iters=2)
        mine.load_state_dict(model.state_dict())
        mine.eval()
    d = {"pc0": batch["pc0"], "pc1": batch["pc1"], "pose0": batch["pose0"], "pose1": batch["pose1"]}
    for k in ("flow", "pose_flow", "pc0_valid_point_idxes", "pc0_points_lst", "pc1_valid_point_idxes", "pc1_points_lst"):
        for b, t in enumerate(res[k]):
            d[f"{k}.{b}"] = t
    # weights: seed 500 + construction order reproduce them; store per-tensor checksums, not 27 MB of floats
    d["seed"] = 500
    for k, v in model.state_dict().items():
        d["wsum." + k] = v.double().sum()
        d["wabs." + k] = v.double().abs().sum()
    np.savez_compressed(os.path.join(OUT, "g5_deflow_orchestration.npz"), **_np(d))
    # variant with explicit ego_motion key [REF deflow.py:64-65]
    batch2 = dict(batch); batch2["ego_motion"] = torch.stack([p[2] for p in pcs])
    with torch.no_grad():
        res2 = model(batch2)
    d2 = {"ego_motion": batch2["ego_motion"]}
    for b, t in enumerate(res2["pose_flow"]):
        d2[f"pose_flow.{b}"] = t
    for b, t in enumerate(res2["flow"]):
        d2[f"flow.{b}"] = t
    np.savez_compressed(os.path.join(OUT, "g5_deflow_egomotion.npz"), **_np(d2))
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
