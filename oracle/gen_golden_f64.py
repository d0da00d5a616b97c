"""float64 twins of the decoder / ConvWithNorms goldens -- TEST INFRASTRUCTURE ONLY, run in the build container.

Run:  python oracle/gen_golden_f64.py        (reads tests/golden/g{2,3,4}_*.npz, writes tests/golden/*_f64.npz)

The committed goldens hold what the REAL reference ([REF decoder.py:72-220], imported unmodified by oracle/gen_golden.py)
computes in fp32.  The north-star tolerance (1e-4) is tighter than two fp32 implementations can be compared at blindly, so
the GPU tests measure both the HIP kernels and the fp32 golden against a third computation: the same reference classes,
same weights, same inputs, executed in float64 (``module.double()``).  Only outputs and gradients are stored.
"""
from __future__ import annotations

import glob
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.gen_golden import OUT, _load_ref_decoder   # noqa: E402


def _t64(a):
    return torch.from_numpy(np.asarray(a)).double()


def _load(mod, g, prefix):
    mod.load_state_dict({k[len(prefix):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(prefix)})
    return mod.double()


def decoder_twin(R, path, cls, **kw):
    g = dict(np.load(path))
    m = _load(cls(**kw), g, "w.")
    before = _t64(g["before"]).requires_grad_(True)
    after = _t64(g["after"]).requires_grad_(True)
    n = sum(1 for k in g if k.startswith("vc"))
    infos = [{"voxel_coords": torch.from_numpy(g[f"vc{i}"]), "point_offsets": _t64(g[f"off{i}"])} for i in range(n)]
    flows = m(before, after, infos)
    sum((f * _t64(g[f"gflow{i}"])).sum() for i, f in enumerate(flows)).backward()
    d = {"gbefore": before.grad, "gafter": after.grad}
    for i, f in enumerate(flows):
        d[f"flow{i}"] = f
    d.update({"gw." + k: p.grad for k, p in m.named_parameters()})
    return {k: v.detach().numpy() for k, v in d.items()}


def cwn_twin(R, path):
    g = dict(np.load(path))
    cin, cout = g["w0.conv.weight"].shape[1], g["w0.conv.weight"].shape[0]
    m = _load(R.ConvWithNorms(cin, cout, int(g["k"]), int(g["s"]), int(g["p"])), g, "w0.")
    m.train(bool(g["train"]))
    x = _t64(g["x"]).requires_grad_(True)
    y = m(x)
    y.backward(_t64(g["gy"]))
    d = {"y": y, "gx": x.grad, "running_mean": m.batchnorm.running_mean, "running_var": m.batchnorm.running_var}
    d.update({"gw." + k: p.grad for k, p in m.named_parameters() if p.grad is not None})
    return {k: v.detach().numpy() for k, v in d.items()}


def main():
    torch.set_num_threads(1)
    R = _load_ref_decoder()
    for it in (1, 4, 8, 16):
        p = os.path.join(OUT, f"g2_grudecoder_it{it}.npz")
        np.savez_compressed(p.replace(".npz", "_f64.npz"), **decoder_twin(R, p, R.ConvGRUDecoder, num_iters=it))
    p = os.path.join(OUT, "g3_lineardecoder.npz")
    np.savez_compressed(p.replace(".npz", "_f64.npz"), **decoder_twin(R, p, R.LinearDecoder))
    for p in sorted(glob.glob(os.path.join(OUT, "g4_convwithnorms_*.npz"))):
        if p.endswith("_f64.npz"):
            continue
        np.savez_compressed(p.replace(".npz", "_f64.npz"), **cwn_twin(R, p))
    print("wrote", sorted(f for f in os.listdir(OUT) if f.endswith("_f64.npz")))


if __name__ == "__main__":
    main()
