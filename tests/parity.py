"""Parity bookkeeping shared by the GPU tests.

The north-star bar is 1e-4 relative on forward AND backward.  Two fp32 implementations of a 40-layer network cannot be
compared with each other at that level blindly -- each carries its own rounding -- so the model-level tests measure both
against a THIRD computation: the oracle re-run in float64 (``copy.deepcopy(ref).double()``, see oracle/ref_torch.py: only the
ego-motion step and the voxel coordinates stay fp32, because they decide which cell a point falls in).  Per tensor

    err(HIP, fp64)  <=  max(1e-4, 4 x err(oracle_fp32, fp64))

in three norms (round 3): max-abs / max|ref| as before, rms-relative (||d||_2 / ||ref||_2: a tensor whose small entries are all
wrong fails this one) and 1 - cosine (bound squared, since 1 - cos ~ rms^2 / 2).

i.e. the HIP path must be within the north-star tolerance of the exact result, or -- for the few tensors where fp32 itself
cannot do better (ill-conditioned sums) -- no worse than a small multiple of what the reference arithmetic achieves.
Every comparison is appended to ``gpurun_out/parity_report.jsonl`` (pulled back by gpurun; also printed with ``-rA``).
"""
import copy
import json
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = os.path.join(ROOT, "gpurun_out", "parity_report.jsonl")
FLOOR, FACTOR = 1e-4, 4.0


def rel_err(got, want) -> float:
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    assert got.shape == want.shape, (got.shape, want.shape)
    assert torch.isfinite(got).all()
    return float((got - want).abs().max() / want.abs().max().clamp_min(1e-30))


def record(test: str, name: str, **kw):
    try:
        os.makedirs(os.path.dirname(REPORT), exist_ok=True)
        with open(REPORT, "a") as f:
            kw = {k: (v.item() if hasattr(v, "item") else v) for k, v in kw.items()}      # numpy / torch scalars
            f.write(json.dumps({"test": test, "tensor": name, **kw}) + "\n")
    except OSError:
        pass


def rms_rel(got, want) -> float:
    """||got - want||_2 / ||want||_2 -- the whole tensor's error, not its single worst element"""
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    return float((got - want).norm() / want.norm().clamp_min(1e-300))


def one_minus_cos(got, want) -> float:
    got, want = got.detach().double().cpu().reshape(-1), want.detach().double().cpu().reshape(-1)
    den = float(got.norm() * want.norm())
    return 0.0 if den == 0.0 else max(0.0, 1.0 - float(torch.dot(got, want)) / den)


def three_way(test: str, name: str, got, ref32, ref64, floor: float = FLOOR, factor: float = FACTOR) -> float:
    """assert err(got, fp64) <= max(floor, factor * err(oracle fp32, fp64)) in THREE norms; returns the max-norm err(got, fp64).
      * max-abs error / max|reference|   (round 1-2's only figure: blind to small entries that are 100 % wrong)
      * rms-relative  ||got - ref||_2 / ||ref||_2   with the same floor / factor
      * 1 - cosine(got, ref) <= max(floor^2, factor^2 * (1 - cos(oracle fp32, ref)))   (direction of the whole tensor; 1 - cos ~ rms^2 / 2)"""
    e_hip, e_o32 = rel_err(got, ref64), rel_err(ref32, ref64)
    r_hip, r_o32 = rms_rel(got, ref64), rms_rel(ref32, ref64)
    c_hip, c_o32 = one_minus_cos(got, ref64), one_minus_cos(ref32, ref64)
    bound, rbound, cbound = max(floor, factor * e_o32), max(floor, factor * r_o32), max(floor * floor, factor * factor * c_o32)
    ok = e_hip <= bound and r_hip <= rbound and c_hip <= cbound
    record(test, name, err_hip_vs_fp64=e_hip, err_oracle32_vs_fp64=e_o32, bound=bound, rms_hip=r_hip, rms_oracle32=r_o32, rms_bound=rbound,
           one_minus_cos_hip=c_hip, one_minus_cos_oracle32=c_o32, cos_bound=cbound, ok=ok)
    print(f"[parity] {test} {name}: HIP vs fp64 max {e_hip:.2e} rms {r_hip:.2e} 1-cos {c_hip:.1e} | oracle fp32 vs fp64 max {e_o32:.2e} "
          f"rms {r_o32:.2e} | bounds {bound:.1e} / {rbound:.1e} / {cbound:.1e}")
    assert e_hip <= bound, f"{test} {name}: err(HIP, fp64) = {e_hip:.3e} > {bound:.1e} (oracle fp32 vs fp64: {e_o32:.3e})"
    assert r_hip <= rbound, f"{test} {name}: rms-rel(HIP, fp64) = {r_hip:.3e} > {rbound:.1e} (oracle fp32 vs fp64: {r_o32:.3e})"
    assert c_hip <= cbound, f"{test} {name}: 1 - cos(HIP, fp64) = {c_hip:.3e} > {cbound:.1e} (oracle fp32 vs fp64: {c_o32:.3e})"
    return e_hip


def is_bn_shadowed_bias(name: str) -> bool:
    """conv biases followed by a training-mode BatchNorm: their true gradient is exactly 0 (fp64 oracle: ~1e-19)"""
    return name.endswith("conv.bias") and "encoder_step" in name


class DyAbsSums:
    """Records sum(|dy|) of every ConvWithNorms backward (dy = gradient at the conv output, after the BatchNorm + GELU
    backward) by wrapping ops.bn_gelu_bwd: the scale against which a BN-shadowed bias gradient -- a sum over dy whose exact
    value is 0 -- has to vanish.  Call order = reverse layer order of the encoder (stage 3 last layer first)."""

    def __init__(self, monkeypatch):
        from deflow_amd import ops
        self.sums = []
        orig = ops.bn_gelu_bwd

        def wrapped(*a, **k):
            out = orig(*a, **k)
            self.sums.append(float(out[0].abs().sum()))
            return out

        monkeypatch.setattr(ops, "bn_gelu_bwd", wrapped)

    def by_module(self, backbone):
        mods = [m for stage in (backbone.encoder_step_1, backbone.encoder_step_2, backbone.encoder_step_3) for m in stage]
        assert len(self.sums) == len(mods), (len(self.sums), len(mods))
        return dict(zip(reversed(mods), self.sums))


def oracle_pair(ref):
    """(fp32 oracle, its float64 twin) -- copy BEFORE either runs (BatchNorm running statistics move)"""
    return ref, copy.deepcopy(ref).double()


def oracle_step(ref, batch, loss_fn: str = "deflowLoss"):
    """forward + loss + backward of an oracle instance -> (result dict, loss, {name: grad})"""
    from oracle import ref_torch as O
    res = ref(batch)
    loss = O.training_loss(res, batch, loss_fn)
    loss.backward()
    return res, loss.detach(), {k: p.grad for k, p in ref.named_parameters()}


def check_step(test: str, mine, res_m, loss_m, o32, o64, dy_sums=None, skip=()):
    """Model-level three-way check of one training step: flow per sample, loss, every parameter gradient.
    o32 / o64 = oracle_step() results.  BN-shadowed conv biases: |grad| <= 1e-6 * sum|dy| (their exact value is 0)."""
    (res32, loss32, g32), (res64, loss64, g64) = o32, o64
    for b in range(len(res64["flow"])):
        if res64["flow"][b].numel():
            three_way(test, f"flow[{b}]", res_m["flow"][b], res32["flow"][b], res64["flow"][b])
    three_way(test, "loss", loss_m.reshape(1), loss32.reshape(1), loss64.reshape(1))
    shadow_scale = dy_sums.by_module(mine.backbone) if dy_sums is not None else None
    worst = 0.0
    for k, p in mine.named_parameters():
        if k in skip:
            continue
        assert p.grad is not None, k
        if is_bn_shadowed_bias(k) and mine.training:
            if shadow_scale is not None:
                mod = dict(mine.backbone.named_modules())[k[len("backbone."):-len(".conv.bias")]]
                bound = 1e-6 * shadow_scale[mod]
                g = float(p.grad.abs().max())
                record(test, "grad " + k, abs_grad=g, bound=bound, sum_abs_dy=shadow_scale[mod], ok=g <= bound)
                assert g <= bound, f"{k}: |grad| {g:.3e} > 1e-6 * sum|dy| = {bound:.3e} (true value: 0)"
            continue
        worst = max(worst, three_way(test, "grad " + k, p.grad, g32[k], g64[k]))
    print(f"[parity] {test}: worst parameter-gradient error vs fp64: {worst:.3e}")
    return worst
