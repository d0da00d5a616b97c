"""Per-tensor DIGESTS of the oracle's B = 16 training step, generated ONCE in the build container (fp32 and float64 oracle; ~20
CPU-minutes on 8 cores at the full size) so that the GPU box checks the three-way parity rule at BASELINE configs[2]'s own size
with no CPU cost (VERDICT r2 next #6b).  Test infrastructure only.

    python oracle/gen_digest_bs16.py 256 20000      -> tests/golden/bs16_256_digest.npz
    python oracle/gen_digest_bs16.py 512 80000      -> tests/golden/bs16_512_digest.npz
    python oracle/gen_digest_bs16.py 1024 160000 --batch 4 --voxel 0.1 --iters 8 --seed 20240116 --init-seed 46 --exact-synth 1
                                                    -> tests/golden/bs4_1024_it8_digest.npz   (BASELINE configs[4] per GPU:
                                                       the shape of bench.py's `configs4_shape`; round 4)

The CPU thread count is PINNED (torch.set_num_threads(8), recorded in the file): the fp32 oracle's sums depend on the
partition of its reductions over threads, and its own error fields (`*.e32_*`) drifted by up to 20 % between regenerations
when the count was left to the machine.

Per parameter gradient (float64 oracle = the yardstick): max|g|, ||g||_2, NPROJ projections on seeded random sign vectors
(an error vector e shows up in a projection as N(0, ||e||_2^2): the projections test the rms error without shipping 27 MB of
gradients), and the fp32 oracle's OWN errors against it (max-abs / rms-relative / 1 - cosine: the `4 x err(oracle fp32)` side of
the rule).  Per sample: the valid-point count, max|flow|, ||flow||_2 and projections of the float64 flow; both losses.
Inputs are regenerated on the GPU box from the same seeds (model init torch.manual_seed(16), synth_batch(16, N, seed=4242))."""
import copy
import os
import sys
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NPROJ = 16


def signs(name: str, n: int, k: int = NPROJ) -> torch.Tensor:
    """[k, n] float64 random +-1, seeded by the tensor's name (CPU generator: identical here and on the GPU box)"""
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    return (torch.randint(0, 2, (k, n), generator=g, dtype=torch.int8).double() * 2.0 - 1.0)


def project(name: str, t: torch.Tensor) -> torch.Tensor:
    v = t.detach().double().reshape(-1).cpu()
    out = torch.empty(NPROJ, dtype=torch.float64)
    for k0 in range(0, NPROJ, 4):          # 4 sign rows at a time: bounded memory for the 2.4 M-element tensors
        out[k0:k0 + 4] = signs(name, v.numel())[k0:k0 + 4] @ v
    return out


class SpillToDisk:
    """autograd saved-tensor hooks that keep every saved activation >= 32 MB in a scratch file instead of in RAM: the float64
    oracle's tape of 16 full-size pairs (2 x 16 images of 512 x 512 x 64..128 channels, conv / BatchNorm / GELU outputs all
    saved) is ~2x the 62 GB of the build container.  Values are written and read back bit for bit: the gradients are those
    of the plain run.  A tensor saved by two nodes is written once (matched by object identity through a weak reference, never
    by address: a freed activation's address is reused by the next one of the same shape)."""
    MIN_BYTES = 1 << 25

    def __init__(self, root):
        import weakref
        self.root, self.n, self.seen, self.weakref, self.bytes = root, 0, {}, weakref, 0
        os.makedirs(root, exist_ok=True)

    def pack(self, t):
        if t.numel() * t.element_size() < self.MIN_BYTES:
            return t
        hit = self.seen.get(id(t))
        if hit is not None and hit[0]() is t and hit[1] == t._version:
            return hit[2]
        path = os.path.join(self.root, f"t{self.n}.bin")
        self.n += 1
        t.detach().contiguous().numpy().tofile(path)
        self.bytes += t.numel() * t.element_size()
        h = (path, t.dtype, tuple(t.shape))
        self.seen[id(t)] = (self.weakref.ref(t), t._version, h)
        return h

    def unpack(self, h):
        if isinstance(h, torch.Tensor):
            return h
        path, dtype, shape = h
        return torch.from_numpy(np.fromfile(path, dtype=torch.empty(0, dtype=dtype).numpy().dtype)).reshape(shape)

    def __enter__(self):
        self.ctx = torch.autograd.graph.saved_tensors_hooks(self.pack, self.unpack)
        self.ctx.__enter__()
        return self

    def __exit__(self, *a):
        import shutil
        self.ctx.__exit__(*a)
        shutil.rmtree(self.root, ignore_errors=True)


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("grid", type=int)
    ap.add_argument("n_pts", type=int)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--voxel", type=float, default=0.2)
    ap.add_argument("--iters", type=int, default=4)
    ap.add_argument("--seed", type=int, default=4242)
    ap.add_argument("--init-seed", type=int, default=16)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--exact-synth", type=int, default=0,
                    help="1: the host-independent generator (deflow_amd.synth exact=True: no BLAS / LAPACK in the input pipeline) -- "
                         "the default generator's pc1 / pose1 / gt flow differ in the last bit between hosts, which is enough to move "
                         "points across 0.1 m voxel edges (found with the configs[4] digest: deep-layer gradients off by percents on "
                         "the GPU box although every kernel was right)")
    ap.add_argument("--weights", default=None,
                    help="a state_dict (torch.save) to take instead of the seeded initialisation -- the CONDITIONED weights of round 5 "
                         "(tests/helpers/conditioned_weights.py: 50 fp32 Adam steps of the HIP trainer, dumped on the GPU box)")
    ap.add_argument("--out", default=None, help="output file name under tests/golden/ (default: derived from the shape)")
    ap.add_argument("--only64", type=int, default=0, help="1: float64 oracle only (no fp32-oracle error fields)")
    args = ap.parse_args()
    grid, n_pts, B = args.grid, args.n_pts, args.batch
    torch.set_num_threads(args.threads)
    from deflow_amd.synth import synth_batch
    from oracle import ref_torch as O
    half = 0.5 * args.voxel * grid
    cfg = dict(voxel_size=[args.voxel, args.voxel, 6], point_cloud_range=[-half, -half, -3, half, half, 3], grid_feature_size=[grid, grid],
               num_iters=args.iters)
    torch.manual_seed(args.init_seed)
    ref = O.DeFlow(**cfg).train()
    sd = copy.deepcopy(ref.state_dict())
    del ref
    if args.weights:
        sd = {k: v.clone() for k, v in torch.load(args.weights, map_location="cpu").items()}
    # (the point spread follows the metric extent of the grid, as synth_batch's grid_hw does for the 0.2 m voxels)
    batch = synth_batch(B, n_pts, seed=args.seed, grid_hw=(int(round(grid * args.voxel / 0.2)),) * 2,
                        exact=bool(args.exact_synth))
    d = {"grid": grid, "n_pts": n_pts, "nproj": NPROJ, "batch": B, "voxel": args.voxel, "iters": args.iters, "seed": args.seed,
         "init_seed": args.init_seed, "threads": args.threads, "exact_synth": args.exact_synth}
    outs = {}
    spill_root = os.environ.get("DF_DIGEST_SPILL", "/tmp/df_digest_spill")
    d["weights"] = os.path.basename(args.weights) if args.weights else ""
    for tag in (("64",) if args.only64 else ("32", "64")):                       # one precision at a time, its tape released before the next
        m = O.DeFlow(**cfg)
        m.load_state_dict(sd)
        m = (m.double() if tag == "64" else m).train()
        with SpillToDisk(os.path.join(spill_root, tag)) as sp:
            res = m(batch)
            loss = O.training_loss(res, batch)
            loss.backward()
            print(tag, "tape spilled:", sp.n, "tensors,", sp.bytes >> 20, "MiB", flush=True)
        outs[tag] = ({"flow": [f.detach().clone() for f in res["flow"]]},
                     {k: p.grad.detach().double() for k, p in m.named_parameters()}, float(loss.detach()))
        del res, loss, m
        print(tag, "loss", outs[tag][2], flush=True)
    if args.only64:
        outs["32"] = outs["64"]          # (error fields of the fp32 oracle read 0: not measured)
    (res32, g32, l32), (res64, g64, l64) = outs["32"], outs["64"]
    d["loss32"], d["loss64"] = l32, l64
    for b in range(B):
        f64, f32 = res64["flow"][b].detach().double(), res32["flow"][b].detach().double()
        d[f"flow.{b}.count"] = f64.shape[0]
        d[f"flow.{b}.max"] = float(f64.abs().max()) if f64.numel() else 0.0
        d[f"flow.{b}.l2"] = float(f64.norm())
        d[f"flow.{b}.proj"] = project(f"flow.{b}", f64).numpy() if f64.numel() else np.zeros(NPROJ)
        d[f"flow.{b}.e32_max"] = float((f32 - f64).abs().max() / f64.abs().max()) if f64.numel() else 0.0
        d[f"flow.{b}.e32_rms"] = float((f32 - f64).norm() / f64.norm()) if f64.numel() else 0.0
    for k, g in g64.items():
        a = g32[k]
        d[f"grad.{k}.max"] = float(g.abs().max())
        d[f"grad.{k}.l2"] = float(g.norm())
        d[f"grad.{k}.proj"] = project("grad." + k, g).numpy()
        d[f"grad.{k}.e32_max"] = float((a - g).abs().max() / g.abs().max().clamp_min(1e-300))
        d[f"grad.{k}.e32_rms"] = float((a - g).norm() / g.norm().clamp_min(1e-300))
        den = float(a.norm() * g.norm())
        d[f"grad.{k}.e32_cos"] = 0.0 if den == 0 else max(0.0, 1.0 - float(torch.dot(a.reshape(-1), g.reshape(-1))) / den)
    name = f"bs16_{grid}_digest.npz" if (B, args.iters) == (16, 4) else f"bs{B}_{grid}_it{args.iters}_digest.npz"
    out = os.path.join(ROOT, "tests", "golden", args.out or name)
    np.savez_compressed(out, **d)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
