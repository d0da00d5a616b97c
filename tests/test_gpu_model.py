"""GPU: the model=deflow plugin end to end against the CPU oracle (same weights, same seeded inputs), the
reference-generated orchestration golden (G5), and size-independent properties at the BASELINE config sizes."""
import math
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SMALL = dict(voxel_size=[0.2, 0.2, 6], point_cloud_range=[-6.4, -6.4, -3, 6.4, 6.4, 3], grid_feature_size=[64, 64])


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda")


def rel_err(got, want):
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    assert got.shape == want.shape, (got.shape, want.shape)
    assert torch.isfinite(got).all()
    return float((got - want).abs().max() / want.abs().max().clamp_min(1e-30))


def check(name, got, want, tol):
    e = rel_err(got, want)
    print(f"[parity] {name}: rel_err={e:.3e} (tol {tol:.0e})")
    assert e <= tol, f"{name}: {e:.3e} > {tol:.1e}"


def make_batch(B, N, seed, grid=64):
    from deflow_amd.synth import synth_pair
    pairs = [synth_pair(seed + b, N, grid_hw=(grid, grid)) for b in range(B)]
    return {"pc0": torch.stack([p[0] for p in pairs]), "pc1": torch.stack([p[1] for p in pairs]),
            "pose0": torch.stack([torch.eye(4) for _ in pairs]),
            "pose1": torch.stack([torch.linalg.inv(p[2]) for p in pairs]),
            "flow": torch.stack([p[3] for p in pairs])}


def to_dev(batch, dev):
    return {k: v.to(dev) for k, v in batch.items()}


def build_pair(dev, seed=0, **kw):
    import deflow_amd
    from oracle import ref_torch as O
    torch.manual_seed(seed)
    ref = O.DeFlow(**SMALL, **kw)
    with torch.no_grad():  # make BN affine / running stats non-trivial
        for m in ref.modules():
            if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                m.weight.uniform_(0.6, 1.4); m.bias.uniform_(-0.2, 0.2)
                m.running_mean.uniform_(-0.3, 0.3); m.running_var.uniform_(0.6, 1.5)
    mine = deflow_amd.DeFlow(**SMALL, **kw)
    mine.load_state_dict(ref.state_dict())
    return ref, mine.to(dev)


@pytest.mark.parametrize("opt", [dict(decoder_option="gru", num_iters=4), dict(decoder_option="linear"),
                                 dict(decoder_option="gru", num_iters=2, align_corners=True)])
def test_forward_eval_vs_oracle(dev, opt):
    ref, mine = build_pair(dev, 1, **opt)
    ref.eval(); mine.eval()
    batch = make_batch(2, 1500, 100)
    with torch.no_grad():
        want = ref(batch)
        got = mine(to_dev(batch, dev))
    for b in range(2):
        assert torch.equal(got["pc0_valid_point_idxes"][b].cpu(), want["pc0_valid_point_idxes"][b])
        assert torch.equal(got["pc1_valid_point_idxes"][b].cpu(), want["pc1_valid_point_idxes"][b])
        assert torch.equal(got["pc1_points_lst"][b].cpu(), want["pc1_points_lst"][b])
        check(f"eval flow b{b} {opt}", got["flow"][b], want["flow"][b], 1e-4)
        m = ~torch.isnan(want["pose_flow"][b])
        check(f"pose_flow b{b}", got["pose_flow"][b].cpu()[m], want["pose_flow"][b][m], 1e-5)


def test_train_step_vs_oracle(dev, monkeypatch):
    """forward in training mode (batch statistics), deflowLoss, backward: flow, loss and EVERY parameter gradient against
    the oracle in fp32 and in fp64 (tests/parity.py: err(HIP, fp64) <= max(1e-4, 4 x err(oracle fp32, fp64))); the 17
    BatchNorm-shadowed conv biases must vanish against sum|dy|; BatchNorm running statistics after the step."""
    from oracle import ref_torch as O
    import parity
    ref, mine = build_pair(dev, 2, decoder_option="gru", num_iters=4)
    ref.train(); mine.train()
    ref, ref64 = parity.oracle_pair(ref)
    batch = make_batch(2, 1500, 200)
    o32, o64 = parity.oracle_step(ref, batch), parity.oracle_step(ref64, batch)
    bd = to_dev(batch, dev)
    sums = parity.DyAbsSums(monkeypatch)
    res_m = mine(bd)
    loss_m = O.training_loss(res_m, bd)  # the trainer's own torch loss on the drop-in result dict
    loss_m.backward()
    parity.check_step("train_step", mine, res_m, loss_m.detach(), o32, o64, dy_sums=sums)
    br = dict(ref.named_buffers())
    for k, v in mine.named_buffers():
        if v.dtype.is_floating_point:
            check(f"buffer {k}", v, br[k], 1e-4)
        else:
            assert int(v) == int(br[k]), k


def test_train_step_scatter_max_vs_oracle(dev):
    """DynamicScatter reduce 'max' (the embedder's mode="max"): forward canvas and the whole training step against the
    oracle (fp32 and fp64), whose backward follows mmcv's rule (the first maximal point of a pillar takes the channel's
    gradient)."""
    from oracle import ref_torch as O
    import parity
    ref, mine = build_pair(dev, 5, decoder_option="gru", num_iters=2)
    ref.embedder.feature_net.mode = "max"
    mine.embedder.mode = 1
    ref.train(); mine.train()
    ref, ref64 = parity.oracle_pair(ref)
    batch = make_batch(2, 3000, 700)       # ~0.7 points per cell on average: many multi-point pillars
    o32, o64 = parity.oracle_step(ref, batch), parity.oracle_step(ref64, batch)
    bd = to_dev(batch, dev)
    res_m = mine(bd)
    loss_m = O.training_loss(res_m, bd)
    loss_m.backward()
    parity.check_step("scatter_max", mine, res_m, loss_m.detach(), o32, o64)
    ref.eval(); mine.eval()
    with torch.no_grad():
        want, got = ref.embedder(batch["pc0"])[0], mine.embedder(bd["pc0"])[0]
    check("max-mode eval canvas", got, want, 1e-5)


def test_full_size_train_step_vs_oracle(dev, monkeypatch):
    """the BASELINE shape itself (512 x 512 grid, 80 000 points per cloud, 4 GRU iterations; one pair): loss, flow and
    every parameter gradient of a training step against the oracle in fp32 and fp64 (about 15 s + 1 min of CPU time).  This
    is the shape at which the 8/12-wave tiles, the two-stage reductions and the sparse edge kernels are actually exercised."""
    import deflow_amd
    import parity
    from oracle import ref_torch as O
    from deflow_amd.synth import synth_batch
    torch.manual_seed(11)
    ref = O.DeFlow()
    mine = deflow_amd.DeFlow()
    mine.load_state_dict(ref.state_dict())
    mine = mine.to(dev)
    ref.train(); mine.train()
    ref, ref64 = parity.oracle_pair(ref)
    batch = synth_batch(1, 80000)
    o32, o64 = parity.oracle_step(ref, batch), parity.oracle_step(ref64, batch)
    bd = to_dev(batch, dev)
    sums = parity.DyAbsSums(monkeypatch)
    res_m = mine(bd)
    loss_m = O.training_loss(res_m, bd)
    loss_m.backward()
    parity.check_step("full_size", mine, res_m, loss_m.detach(), o32, o64, dy_sums=sums)


def _bs16_case(dev, monkeypatch, grid, n_pts, tag, golden_dir, digest=None, bf16=None):
    """B = 16 training step through the bench's own path (Trainer: forward_padded, fused loss kernel, hand-sequenced backward
    into the gradient arena) against the oracle's DIGESTS (oracle/gen_digest_bs16.py, generated once in the build container from
    the fp32 AND float64 oracle -- the float64 twin of 16 full-size pairs costs ~10 CPU-minutes, which the GPU box no longer
    pays): per gradient ||g||_2, 16 seeded random-sign projections and the fp32 oracle's own errors.  The three-way rule per
    tensor in its rms form:  a projection of the error vector is N(0, ||e||_2^2), so
        |proj(HIP) - proj(fp64)| <= 4.5 x max(1e-4, 4 x rms_err(oracle fp32)) x ||g_fp64||_2     for all 16 projections
    (4.5 sigma over 16 x ~110 draws), plus ||g||_2 itself within that relative bound; loss and per-sample flows likewise."""
    import deflow_amd
    import parity
    from oracle.gen_digest_bs16 import project
    from deflow_amd.synth import synth_batch
    dg = dict(np.load(os.path.join(golden_dir, digest or f"bs16_{grid}_digest.npz")))
    assert int(dg["grid"]) == grid and int(dg["n_pts"]) == n_pts
    # (digests of round 3 carry no generator arguments: B = 16, 0.2 m voxels, 4 iterations, seeds 16 / 4242)
    B = int(dg["batch"]) if "batch" in dg else 16
    voxel = float(dg["voxel"]) if "voxel" in dg else 0.2
    iters = int(dg["iters"]) if "iters" in dg else 4
    half = 0.5 * voxel * grid
    cfg = dict(voxel_size=[voxel, voxel, 6], point_cloud_range=[-half, -half, -3, half, half, 3], grid_feature_size=[grid, grid],
               num_iters=iters)
    from oracle import ref_torch as O
    torch.manual_seed(int(dg["init_seed"]) if "init_seed" in dg else 16)
    ref = O.DeFlow(**cfg)                       # (only for its seeded initial weights: the oracle does not run here)
    mine = deflow_amd.DeFlow(**cfg)
    mine.load_state_dict(ref.state_dict())
    mine = mine.to(dev).train()
    batch = synth_batch(B, n_pts, seed=int(dg["seed"]) if "seed" in dg else 4242, grid_hw=(int(round(grid * voxel / 0.2)),) * 2,
                        exact=bool(int(dg["exact_synth"])) if "exact_synth" in dg else False)
    bd = to_dev(batch, dev)
    sums = parity.DyAbsSums(monkeypatch)
    from deflow_amd.optim import Trainer
    tr = Trainer(mine, lr=0.0)
    tr.flat.zero_grad(); tr.sink.begin()
    mine.forward_padded(bd)
    loss_m = tr.loss_on_last_forward(bd)
    loss_m.backward()
    torch.cuda.synchronize()
    st = mine.last_state
    m0 = st["counts0"].tolist()
    assert abs(float(loss_m.detach()) - float(dg["loss64"])) <= max(1e-4, 4 * abs(float(dg["loss32"]) - float(dg["loss64"])) / abs(float(dg["loss64"]))) * abs(float(dg["loss64"]))
    SIG = 4.5
    for b in range(B):
        assert m0[b] == int(dg[f"flow.{b}.count"]), b
        if m0[b]:
            bound = max(1e-4, 4 * float(dg[f"flow.{b}.e32_rms"])) * float(dg[f"flow.{b}.l2"])
            dp = (project(f"flow.{b}", st["flow"][b, :m0[b]]).numpy() - dg[f"flow.{b}.proj"])
            assert np.abs(dp).max() <= SIG * bound, (b, float(np.abs(dp).max()), bound)
    shadow = sums.by_module(mine.backbone)
    mods = dict(mine.backbone.named_modules())
    worst = (0.0, "")
    bad = []
    for k, p in mine.named_parameters():
        if parity.is_bn_shadowed_bias(k):
            sc = shadow[mods[k[len("backbone."):-len(".conv.bias")]]]
            assert float(p.grad.abs().max()) <= 1e-6 * sc, k
            continue
        l2 = float(dg[f"grad.{k}.l2"])
        rel = max(1e-4, 4 * float(dg[f"grad.{k}.e32_rms"]))
        dp = np.abs(project("grad." + k, p.grad).numpy() - dg[f"grad.{k}.proj"]).max()
        dn = abs(float(p.grad.double().norm()) - l2)
        est = dp / max(l2, 1e-300)          # ~ a few sigma of the rms-relative error
        parity.record(tag, "grad " + k, max_proj_err_over_l2=est, rms_bound=rel, oracle32_rms=float(dg[f"grad.{k}.e32_rms"]), ok=dp <= SIG * rel * l2)
        worst = max(worst, (est / rel, k))
        if not (dp <= SIG * rel * l2 and dn <= SIG * rel * l2):
            bad.append((k, dp / max(l2, 1e-300), dn / max(l2, 1e-300), rel))
    assert not bad, f"{len(bad)} gradients outside the bound (name, projection error / ||g||, norm error / ||g||, rms bound): {bad[:8]}"
    print(f"[parity] {tag}: worst projection error / (bound x ||g||) = {worst[0]:.2f} sigma-units of {SIG} allowed ({worst[1]})")
    if grid == 256 if bf16 is None else bf16:
        _bf16_step_vs_digest(dev, cfg, ref, bd, dg, tag)


BF16_GRAD_RMS = 1e-1      # stated bound of the bf16 training mode: per-tensor rms-relative error of every parameter gradient


def _bf16_step_vs_digest(dev, cfg, ref, bd, dg, tag):
    """the bf16 training mode (bf16 MFMA operands everywhere + bf16 STORAGE of the UNet's activations and
    gradients) against the ORACLE, not against the fp32 HIP step: every parameter gradient of one B = 16 step vs the oracle's
    float64 gradients through their digests -- 16 random-sign projections per tensor, each an N(0, ||e||_2^2) draw of the error
    vector: |proj(bf16) - proj(fp64)| <= 4.5 x 1e-1 x ||g||_2 (stated bound: rms-relative 1e-1; measured directly against the full
    float64 gradients in round 3: worst 7.7e-2, 1 - cosine 3.0e-3, both on the stride-2 first conv of encoder stage 2), the
    gradient norm within 1e-1, the loss within 2e-3, per-sample flow projections within 2e-2 of ||flow||_2."""
    import deflow_amd
    import parity
    from oracle.gen_digest_bs16 import project
    from deflow_amd.optim import Trainer
    from deflow_amd import ops
    m16 = deflow_amd.DeFlow(**cfg)
    m16.load_state_dict(ref.state_dict())
    m16 = m16.to(dev).train()
    tr = Trainer(m16, lr=0.0, dtype="bf16")
    assert tr.bf16_store
    with ops.mfma_bf16(True, True), torch.no_grad():
        tr.flat.zero_grad(); tr.sink.begin()
        loss16 = tr._forward_backward(bd)
    torch.cuda.synchronize()
    loss64 = float(dg["loss64"])
    assert abs(float(loss16) - loss64) <= 2e-3 * abs(loss64), (float(loss16), loss64)
    st = m16.last_state
    m0 = st["counts0"].tolist()
    for b in range(len(m0)):
        if m0[b]:
            dp = np.abs(project(f"flow.{b}", st["flow"][b, :m0[b]]).numpy() - dg[f"flow.{b}.proj"]).max()
            assert dp <= 4.5 * 2e-2 * float(dg[f"flow.{b}.l2"]), (b, dp)
    worst = (0.0, "")
    for k, p in m16.named_parameters():
        if parity.is_bn_shadowed_bias(k):
            continue
        l2 = float(dg[f"grad.{k}.l2"])
        dp = np.abs(project("grad." + k, p.grad).numpy() - dg[f"grad.{k}.proj"]).max()
        dn = abs(float(p.grad.double().norm()) - l2)
        parity.record(tag + "_bf16", "grad " + k, max_proj_err_over_l2=dp / l2, norm_err=dn / l2, rms_bound=BF16_GRAD_RMS,
                      ok=dp <= 4.5 * BF16_GRAD_RMS * l2)
        worst = max(worst, (dp / l2, k))
        assert dp <= 4.5 * BF16_GRAD_RMS * l2 and dn <= BF16_GRAD_RMS * l2, (k, dp / l2, dn / l2)
    print(f"[parity] {tag} bf16 mode vs float64 oracle digests: loss {float(loss16):.6f} / {loss64:.6f}; worst projection error / ||g|| "
          f"{worst[0]:.3e} ({worst[1]}) -- bound 4.5 x {BF16_GRAD_RMS}")


def test_bs16_train_step_vs_oracle(dev, monkeypatch, golden_dir):
    """BASELINE configs[2] is quoted at bs = 16: the batch size changes the split-K partitions of the weight gradients, the
    two-stage BatchNorm reductions (thousands of tile partials per group), the max(1, 256 // B) block counts and work lists
    of the sparse edge kernels.  16 pairs on a 256 x 256 grid / 20 000 points (every bs-dependent branch, seconds of CPU)
    through the bench's own path (Trainer: fused loss kernel, gradient arena) against the fp32 / float64 oracle digests; then the
    bf16 training mode against the same float64 digests."""
    _bs16_case(dev, monkeypatch, 256, 20000, "bs16_256", golden_dir)


def test_bs16_full_size_train_step_vs_oracle(dev, monkeypatch, golden_dir):
    """configs[2] EXACTLY: 16 pairs x 80 000 points on the 512 x 512 grid, one training step against the digests of the fp32 and
    float64 oracle generated in the build container (round 2 could only afford the fp32 oracle here, at twice the bound)."""
    _bs16_case(dev, monkeypatch, 512, 80000, "bs16_512", golden_dir)


def test_bs16_full_size_train_step_gru_fp32_vs_oracle(dev, monkeypatch, golden_dir):
    """the same configs[2] step with the GRU decoder's GEMMs back on the fp32 MFMA (DF_GRU_X2=0): the second row of the bench
    line's `model_error_budget` (what the 16-significant-bit bf16x2 decoder spends of the 1e-4 bound)"""
    monkeypatch.setenv("DF_GRU_X2", "0")
    _bs16_case(dev, monkeypatch, 512, 80000, "bs16_512_gru_fp32", golden_dir)


def test_fastflow3d_train_step_vs_oracle(dev):
    """decoder_option=linear (the fastflow3d head): training step gradients vs the oracle (fp32 and fp64)"""
    from oracle import ref_torch as O
    import parity
    ref, mine = build_pair(dev, 4, decoder_option="linear")
    ref.train(); mine.train()
    ref, ref64 = parity.oracle_pair(ref)
    batch = make_batch(2, 1200, 400)
    o32, o64 = parity.oracle_step(ref, batch), parity.oracle_step(ref64, batch)
    bd = to_dev(batch, dev)
    res_m = mine(bd)
    loss_m = O.training_loss(res_m, bd)
    loss_m.backward()
    parity.check_step("fastflow3d", mine, res_m, loss_m.detach(), o32, o64)


def test_fastflow3d_voxel04_train_step_vs_oracle(dev):
    """the reference's fastflow3d ablation geometry [REF assets/slurm/1_train.sh:78; README.md:68]: voxel_size=[0.4, 0.4, 6] over the
    full +-51.2 m range = a 256 x 256 canvas, LinearDecoder head, one B = 1 training step vs the oracle in fp32 and float64"""
    import deflow_amd
    from oracle import ref_torch as O
    import parity
    cfg = dict(voxel_size=[0.4, 0.4, 6], point_cloud_range=[-51.2, -51.2, -3, 51.2, 51.2, 3], grid_feature_size=[256, 256],
               decoder_option="linear")
    torch.manual_seed(40)
    ref = O.DeFlow(**cfg).train()
    mine = deflow_amd.DeFlow(**cfg)
    mine.load_state_dict(ref.state_dict())
    mine = mine.to(dev).train()
    ref, ref64 = parity.oracle_pair(ref)
    from deflow_amd.synth import synth_batch
    batch = synth_batch(1, 30000, seed=404)                # AV2-shaped cloud over the full range (0.4 m pillars)
    o32, o64 = parity.oracle_step(ref, batch), parity.oracle_step(ref64, batch)
    bd = to_dev(batch, dev)
    res_m = mine(bd)
    loss_m = O.training_loss(res_m, bd)
    loss_m.backward()
    parity.check_step("fastflow3d_voxel04", mine, res_m, loss_m.detach(), o32, o64)


def test_eval_mode_backward_vs_oracle(dev):
    """model.eval() with autograd recording (fine-tuning with frozen BatchNorm, saliency, gradient checks): the reference
    nn.Module is differentiable in eval mode, so is this one -- running statistics in the forward, no batch-statistic terms
    in the backward, conv biases no longer shadowed (every one of them is checked), running statistics untouched."""
    from oracle import ref_torch as O
    import parity
    ref, mine = build_pair(dev, 6, decoder_option="gru", num_iters=3)
    ref.eval(); mine.eval()
    ref, ref64 = parity.oracle_pair(ref)
    buf0 = {k: v.clone() for k, v in mine.named_buffers()}
    batch = make_batch(2, 1500, 900)
    o32, o64 = parity.oracle_step(ref, batch), parity.oracle_step(ref64, batch)
    bd = to_dev(batch, dev)
    res_m = mine(bd)
    assert res_m["flow"][0].grad_fn is not None, "eval-mode forward under autograd must be differentiable"
    loss_m = O.training_loss(res_m, bd)
    loss_m.backward()
    parity.check_step("eval_backward", mine, res_m, loss_m.detach(), o32, o64)
    for k, v in mine.named_buffers():
        assert torch.equal(v, buf0[k]), f"eval mode must not touch {k}"
    # and without autograd the same call takes the tape-less inference path with identical values
    with torch.no_grad():
        res_n = mine(bd)
    for b in range(2):
        check(f"eval no_grad flow[{b}]", res_n["flow"][b], res_m["flow"][b], 1e-5)


def test_fused_loss_path_matches_list_path(dev):
    """the padded fast path (DeflowLossFn on last_state) == the drop-in list path"""
    from deflow_amd.autograd import DeflowLossFn
    from deflow_amd._lib import call, ptr, stream
    from oracle import ref_torch as O
    _, mine = build_pair(dev, 3, decoder_option="gru", num_iters=2)
    mine.train()
    bd = to_dev(make_batch(2, 1200, 300), dev)
    res = mine(bd)
    l1 = O.training_loss(res, bd)
    st = mine.last_state
    B, N, _ = st["flow"].shape
    gt = torch.empty(B, N, 3, device=dev)
    call("df_gather_gt", ptr(bd["flow"].contiguous()), ptr(st["pose_flow"]), ptr(st["idx_c0"]), ptr(st["counts0"]), B, N,
         ptr(gt), 8, stream())
    l2 = DeflowLossFn.apply(st["flow"], gt, st["counts0"])
    check("fused loss", l2.reshape(1), l1.reshape(1), 1e-5)
    g1 = torch.autograd.grad(l1, st["flow"], retain_graph=True)[0]
    g2 = torch.autograd.grad(l2, st["flow"])[0]
    for b in range(B):
        c = int(st["counts0"][b])
        check(f"fused loss grad b{b}", g2[b, :c], g1[b, :c], 1e-5)


def test_g5_orchestration_golden(dev, golden_dir):
    """reference deflow.py executed in the build container (oracle/gen_golden.py): result-dict contract"""
    import deflow_amd
    from oracle import ref_torch as O
    g = dict(np.load(os.path.join(golden_dir, "g5_deflow_orchestration.npz")))
    torch.manual_seed(int(g["seed"]))
    ref = O.DeFlow(**SMALL, decoder_option="gru", num_iters=2)
    mine = deflow_amd.DeFlow(**SMALL, decoder_option="gru", num_iters=2)
    mine.load_state_dict(ref.state_dict())
    mine = mine.to(dev).eval()
    batch = {k: torch.from_numpy(g[k]).to(dev) for k in ("pc0", "pc1", "pose0", "pose1")}
    with torch.no_grad():
        res = mine(batch)
    assert set(res) == {"flow", "pose_flow", "pc0_valid_point_idxes", "pc0_points_lst", "pc1_valid_point_idxes", "pc1_points_lst"}
    for b in range(2):
        assert torch.equal(res["pc0_valid_point_idxes"][b].cpu(), torch.from_numpy(g[f"pc0_valid_point_idxes.{b}"]))
        assert torch.equal(res["pc1_valid_point_idxes"][b].cpu(), torch.from_numpy(g[f"pc1_valid_point_idxes.{b}"]))
        check(f"g5 pc0_points b{b}", res["pc0_points_lst"][b], torch.from_numpy(g[f"pc0_points_lst.{b}"]), 1e-5)
        check(f"g5 flow b{b}", res["flow"][b], torch.from_numpy(g[f"flow.{b}"]), 1e-4)
        w = torch.from_numpy(g[f"pose_flow.{b}"])
        m = ~torch.isnan(w)
        check(f"g5 pose_flow b{b}", res["pose_flow"][b].cpu()[m], w[m], 1e-4)


@pytest.mark.parametrize("form", ["rigid", "general"])
def test_pose_inverse_forms_vs_oracle(dev, monkeypatch, form):
    """both restatements of the UNPINNED upstream helper cal_pose0to1 (closed-form rigid inverse = default; torch.linalg.inv =
    rounds 1-2) through the whole forward with NON-identity pose0 and pose1: pose_flow, the transformed points and the flow
    against the oracle using the same form"""
    from oracle import ref_torch as O
    from deflow_amd import deflow as D
    monkeypatch.setattr(O, "POSE_INVERSE", form)
    monkeypatch.setattr(D, "POSE_INVERSE", form)
    ref, mine = build_pair(dev, 3, decoder_option="gru", num_iters=2)
    ref.eval(); mine.eval()
    batch = make_batch(2, 1500, 4100)
    g = torch.Generator().manual_seed(8)
    world = []
    for b in range(2):            # world poses: pose1 = W, pose0 = W @ T  ->  inv(pose1) @ pose0 = T up to rounding
        yaw = float(torch.rand(1, generator=g)) * 6.28
        W = torch.eye(4)
        W[0, 0] = W[1, 1] = math.cos(yaw); W[0, 1] = -math.sin(yaw); W[1, 0] = math.sin(yaw)
        W[:3, 3] = torch.randn(3, generator=g) * 3.0
        world.append(W)
    T = torch.stack([torch.linalg.inv(p) for p in batch["pose1"]])
    batch["pose1"] = torch.stack(world)
    batch["pose0"] = torch.stack([w @ t for w, t in zip(world, T)])
    with torch.no_grad():
        want = ref(batch)
        got = mine(to_dev(batch, dev))
    for b in range(2):
        w = want["pose_flow"][b]
        m = ~torch.isnan(w)
        check(f"pose_flow[{b}] ({form})", got["pose_flow"][b].cpu()[m], w[m], 1e-5)
        # a 4x4 product on the device vs on the host may differ in the last bit, which can move a point sitting on a cell
        # edge: compare the points both sides kept
        gi, wi = got["pc0_valid_point_idxes"][b].cpu(), want["pc0_valid_point_idxes"][b]
        common = torch.isin(gi, wi)
        assert common.float().mean() > 0.995 and abs(len(gi) - len(wi)) <= 3
        sel_w = torch.isin(wi, gi)
        tol = 1e-4 if (bool(common.all()) and len(gi) == len(wi)) else 5e-3   # a moved point changes its two pillars' features
        check(f"flow[{b}] ({form})", got["flow"][b].cpu()[common], want["flow"][b][sel_w], tol)


def test_full_size_properties(dev):
    """BASELINE config 2 shape (512x512, 80k points, 4 iterations): size-independent properties."""
    import deflow_amd
    torch.manual_seed(0)
    m = deflow_amd.DeFlow().to(dev).eval()
    from deflow_amd.synth import synth_pair
    p = synth_pair(20240116, 80000)
    batch = {"pc0": p[0][None].to(dev), "pc1": p[1][None].to(dev), "pose0": torch.eye(4)[None].to(dev),
             "pose1": torch.linalg.inv(p[2])[None].to(dev)}
    with torch.no_grad():
        r1 = m(batch)
        r2 = m(batch)
    f = r1["flow"][0]
    assert torch.isfinite(f).all() and f.shape[1] == 3
    assert torch.equal(r1["flow"][0], r2["flow"][0]), "the engine is deterministic (no float atomics)"
    idx = r1["pc0_valid_point_idxes"][0]
    assert (idx[1:] > idx[:-1]).all(), "valid indices are strictly increasing (stable compaction)"
    pts = r1["pc0_points_lst"][0]
    assert f.shape[0] == idx.shape[0] == pts.shape[0] and not torch.isnan(pts).any()
    assert (pts[:, :2].abs() <= 51.2).all() and (pts[:, 2] >= -3).all() and (pts[:, 2] < 3).all()
    # permuting the input points permutes the output rows (pillar sums are order-independent up to fp32 rounding)
    perm = torch.randperm(80000, device=dev)
    b2 = dict(batch); b2["pc0"] = batch["pc0"][:, perm]
    with torch.no_grad():
        r3 = m(b2)
    inv = torch.empty_like(perm); inv[perm] = torch.arange(80000, device=dev)
    orig_of_r3 = perm[r3["pc0_valid_point_idxes"][0]]
    order = torch.argsort(orig_of_r3)
    assert torch.equal(orig_of_r3[order], idx)
    e = float((r3["flow"][0][order] - f).abs().max() / f.abs().max())
    print(f"[property] permutation equivariance rel diff {e:.3e}")
    assert e < 1e-3


def test_trainer_learns_and_cli_runs(dev, tmp_path):
    """rows N1/N3: the flat-arena Adam trainer reduces deflowLoss on a fixed batch; the key=value CLI runs an epoch,
    evaluates EPE and writes a checkpoint the reference-style loader accepts."""
    import deflow_amd
    from deflow_amd.optim import Trainer
    from deflow_amd.synth import synth_batch
    from deflow_amd import train as T
    from oracle import ref_torch as O
    ref, m = build_pair(dev, 7, decoder_option="gru", num_iters=2)
    ref.train(); m.train()
    tr = Trainer(m, lr=2e-4)
    opt = torch.optim.Adam(ref.parameters(), lr=2e-4)
    batch_cpu = make_batch(2, 1500, 500)
    batch = to_dev(batch_cpu, dev)
    got, want = [], []
    for _ in range(6):                      # same data every step: trajectories must coincide step by step
        got.append(float(tr.step(batch)))
        opt.zero_grad()
        l = O.training_loss(ref(batch_cpu), batch_cpu)
        l.backward()
        opt.step()
        want.append(float(l))
    print("[train] HIP trainer losses", [round(x, 5) for x in got])
    print("[train] oracle+Adam losses", [round(x, 5) for x in want])
    for g_, w_ in zip(got, want):
        assert abs(g_ - w_) <= 2e-3 * abs(w_), (got, want)
    pr = dict(ref.named_parameters())
    worst = max(rel_err(p_, pr[k]) for k, p_ in m.named_parameters() if not (k.endswith("conv.bias") and "encoder_step" in k))
    print(f"[train] worst parameter rel diff after 6 Adam steps: {worst:.3e}")
    assert worst < 5e-3
    ck = os.path.join(tmp_path, "cli.ckpt")
    T.main(["model=deflow", "lr=2e-4", "epochs=1", "batch_size=2", "loss_fn=deflowLoss", "model.target.num_iters=2",
            "voxel_size=[0.2, 0.2, 6]", "point_cloud_range=[-6.4, -6.4, -3, 6.4, 6.4, 3]", "pairs_per_epoch=4",
            "points_per_cloud=1200", f"save_checkpoint={ck}"])
    m2 = deflow_amd.DeFlow(**SMALL, num_iters=2)
    r = m2.load_from_checkpoint(ck)
    assert not r.missing_keys and not r.unexpected_keys
    # the same command with the step captured as a HIP graph and bf16 MFMA operands (synthetic batches have constant shapes)
    ck2 = os.path.join(tmp_path, "cli_graph.ckpt")
    T.main(["model=deflow", "lr=2e-4", "epochs=1", "batch_size=2", "loss_fn=deflowLoss", "model.target.num_iters=2",
            "voxel_size=[0.2, 0.2, 6]", "point_cloud_range=[-6.4, -6.4, -3, 6.4, 6.4, 3]", "pairs_per_epoch=8",
            "points_per_cloud=1200", "dtype=bf16", "graph=true", "log_every=1", f"save_checkpoint={ck2}"])
    sd = torch.load(ck2, map_location="cpu", weights_only=False)
    # capture's warm-up launches are not training steps (state restored): Adam's step count equals the global step
    assert sd["global_step"] == 4 and sd["optimizer_states"][0]["step"] == 4
    assert all(torch.isfinite(v).all() for v in sd["state_dict"].values() if v.dtype.is_floating_point)


def test_fp32_trajectory_on_the_16bit_pipe_kernels(dev):
    """six Adam steps of the DEFAULT fp32 trainer on a 256 x 256 grid -- large enough that the 3x3 convolutions and weight gradients
    run as the fp16x2 kernels and the GRU decoder as bf16x2 (asserted by kernel name) -- against the fp32 oracle stepping
    torch.optim.Adam on the same batches: the loss trajectory within 1e-3 relative step by step (the small-grid trajectory test
    above never reaches these kernels: its layers are narrower than their tiles), the loss must go down"""
    import deflow_amd
    from oracle import ref_torch as O
    from deflow_amd import ops
    from deflow_amd.optim import Trainer
    from deflow_amd.synth import synth_batch
    cfg = dict(voxel_size=[0.4, 0.4, 6], point_cloud_range=[-51.2, -51.2, -3, 51.2, 51.2, 3], grid_feature_size=[256, 256], num_iters=2)
    torch.manual_seed(77)
    ref = O.DeFlow(**cfg).train()
    mine = deflow_amd.DeFlow(**cfg)
    mine.load_state_dict(ref.state_dict())
    mine = mine.to(dev).train()
    tr = Trainer(mine, lr=2e-4)
    opt = torch.optim.Adam(ref.parameters(), lr=2e-4)
    batches = [synth_batch(2, 20000, seed=900 + 7 * i) for i in range(2)]
    got, want = [], []
    for i in range(6):
        b = batches[i % 2]
        if i == 0:
            ops.PROFILER = prof = ops.KernelProfiler()
        try:
            got.append(float(tr.step(to_dev(b, dev))))
        finally:
            ops.PROFILER = None
        opt.zero_grad()
        l = O.training_loss(ref(b), b)
        l.backward()
        opt.step()
        want.append(float(l.detach()))
    names = {r[0] for r in prof.records}
    assert any(n.startswith("conv_halo_x3_kernel") and n.endswith(",2>") for n in names) and "wgrad3_x3_kernel<2>" in names, names
    assert deflow_amd.decoder.ConvGRUDecoder._x2_on()
    print("[parity] fp32 trajectory on the 16-bit-pipe kernels:", [f"{g:.5f}/{w:.5f}" for g, w in zip(got, want)])
    import parity
    for i, (g, w) in enumerate(zip(got, want)):
        e = abs(g - w) / abs(w)
        parity.record("fp32_16bit_pipe_train", f"loss step {i}", err_vs_fp32_oracle=e, bound=1e-3, ok=e <= 1e-3)
        assert e <= 1e-3, (i, g, w)
    assert got[4] < got[0] and got[5] < got[1]


def test_edge_cases_empty_ragged_and_big_grid(dev):
    """SURVEY 8(c) edge cases: a sample with no valid point, pc0/pc1 padded to different lengths, and the 1024x1024 grid
    (BASELINE config 5 shape, voxel 0.1 m, 160k points) through the whole engine."""
    import deflow_amd
    from oracle import ref_torch as O
    ref, mine = build_pair(dev, 9, decoder_option="gru", num_iters=2)
    ref.eval(); mine.eval()
    batch = make_batch(3, 1500, 600)
    batch["pc1"] = batch["pc1"][:, :1100].contiguous()          # ragged: N' != N
    batch["pc0"][1] = float("nan")                               # sample 1: nothing valid in pc0
    with torch.no_grad():
        want = ref(batch)
        got = mine(to_dev(batch, dev))
    assert got["flow"][1].shape == (0, 3) and got["pc0_valid_point_idxes"][1].numel() == 0
    for b in (0, 2):
        assert torch.equal(got["pc0_valid_point_idxes"][b].cpu(), want["pc0_valid_point_idxes"][b])
        assert torch.equal(got["pc1_valid_point_idxes"][b].cpu(), want["pc1_valid_point_idxes"][b])
        check(f"ragged flow b{b}", got["flow"][b], want["flow"][b], 1e-4)
    # 1024 x 1024 grid, 160k points: finite, deterministic, correctly sized
    torch.manual_seed(0)
    big = deflow_amd.DeFlow(voxel_size=[0.1, 0.1, 6], grid_feature_size=[1024, 1024], num_iters=8).to(dev).eval()
    from deflow_amd.synth import synth_pair
    p = synth_pair(77, 160000)
    bb = {"pc0": p[0][None].to(dev), "pc1": p[1][None].to(dev), "pose0": torch.eye(4)[None].to(dev),
          "pose1": torch.linalg.inv(p[2])[None].to(dev)}
    with torch.no_grad():
        r1, r2 = big(bb), big(bb)
    f = r1["flow"][0]
    assert torch.isfinite(f).all() and f.shape[0] == r1["pc0_valid_point_idxes"][0].numel() > 100000
    assert torch.equal(f, r2["flow"][0])


def test_bf16_inference_path_vs_oracle(dev):
    """BASELINE configs[4] ("bf16 MFMA"): eval-mode forward with the UNet on v_mfma_f32_32x32x16_bf16 (bf16 activations and
    weights, fp32 accumulation / BatchNorm / GELU; pillars and GRU decoder fp32) against the fp32 CPU oracle.
    Stated tolerance: 2e-2 of the largest flow component (bf16 carries 8 mantissa bits through ~30 conv layers;
    measured 3e-4 .. 1e-3), integer outputs still bit-exact."""
    ref, mine = build_pair(dev, 21, decoder_option="gru", num_iters=4)
    ref.eval(); mine.eval()
    mine.inference_dtype = "bf16"
    batch = make_batch(2, 1500, 300)
    with torch.no_grad():
        want = ref(batch)
        got = mine(to_dev(batch, dev))
    for b in range(2):
        assert torch.equal(got["pc0_valid_point_idxes"][b].cpu(), want["pc0_valid_point_idxes"][b])
        w, g = want["flow"][b], got["flow"][b].cpu()
        err = float((g - w).abs().max() / w.abs().max())
        print(f"[parity] bf16 inference flow b{b}: max abs err / max |flow| = {err:.2e} (tol 2e-2)")
        assert err <= 2e-2
    # the switch only affects eval-mode forwards: a training step still runs (and matches) the fp32 kernels
    mine.train(); ref.train()
    from oracle import ref_torch as O
    lr_ = O.training_loss(ref(batch), batch)
    lm_ = O.training_loss(mine(to_dev(batch, dev)), to_dev(batch, dev))
    check("loss with bf16 inference switch set", lm_.reshape(1), lr_.reshape(1), 1e-4)


def _bf16_flow_err(got, want):
    g, w = got.detach().float().cpu(), want.detach().float()
    return float((g - w).abs().max() / w.abs().max()), float(((g - w) ** 2).mean().sqrt() / (w ** 2).mean().sqrt())


def test_bf16_configs4_shape_vs_oracle(dev):
    """BASELINE configs[4] at its own shape: 1024 x 1024 grid (voxel 0.1 m), 160 000 points per cloud, 8 GRU iterations,
    bf16 MFMA -- one pair, eval-mode forward (fp32 HIP path AND bf16 path) against the fp32 CPU oracle.  At this size the
    rolling-row 64->64 kernel (>= 8192 row tiles), the haloed bf16 tiles and the zero-padded first conv (reads 32 channels
    past each cloud's half of the canvas) are all on the path.  fp32: 1e-4; bf16: 2e-2 of the largest flow component (8
    mantissa bits through ~30 layers + 8 GRU iterations), rms error reported."""
    import deflow_amd
    from oracle import ref_torch as O
    from deflow_amd.synth import synth_pair
    cfg = dict(voxel_size=[0.1, 0.1, 6], grid_feature_size=[1024, 1024], num_iters=8)
    torch.manual_seed(44)
    ref = O.DeFlow(**cfg)
    with torch.no_grad():   # non-trivial BatchNorm state, as a trained checkpoint would have
        for m in ref.modules():
            if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                m.weight.uniform_(0.6, 1.4); m.bias.uniform_(-0.2, 0.2)
                m.running_mean.uniform_(-0.3, 0.3); m.running_var.uniform_(0.6, 1.5)
    mine = deflow_amd.DeFlow(**cfg)
    mine.load_state_dict(ref.state_dict())
    mine = mine.to(dev).eval()
    ref.eval()
    p = synth_pair(77, 160000)
    # ego_motion given directly: inverting pose1 on the host vs on the device differs in the last bit, and with 160 000
    # points on 0.1 m cells that is enough to move a point across a cell edge
    batch = {"pc0": p[0][None], "pc1": p[1][None], "pose0": torch.eye(4)[None], "pose1": torch.linalg.inv(p[2])[None],
             "ego_motion": p[2][None]}
    with torch.no_grad():
        want = ref(batch)
        got32 = mine(to_dev(batch, dev))
        mine.inference_dtype = "bf16"
        got16 = mine(to_dev(batch, dev))
    assert torch.equal(got16["pc0_valid_point_idxes"][0].cpu(), want["pc0_valid_point_idxes"][0])
    check("configs[4] shape fp32 flow", got32["flow"][0], want["flow"][0], 1e-4)
    emax, erms = _bf16_flow_err(got16["flow"][0], want["flow"][0])
    import parity
    parity.record("bf16_cfg4", "flow", max_err_over_max_flow=emax, rms_err_over_rms_flow=erms, bound=2e-2, ok=emax <= 2e-2)
    print(f"[parity] bf16 @1024x1024/160k/8 iters: max err / max|flow| = {emax:.2e}, rms err / rms flow = {erms:.2e} (tol 2e-2)")
    assert emax <= 2e-2


def test_bf16_bs16_vs_fp32_and_oracle(dev):
    """bf16 inference at B = 16 on the 512 x 512 grid (the encoder's 64->64 layers then reach the rolling-row kernel: 32
    images x 4 segments x 256 rows): every sample against the fp32 HIP forward of the same batch, samples 0 and 15 also
    against the CPU oracle (eval mode has no cross-sample coupling, so the oracle runs them alone)."""
    import deflow_amd
    from oracle import ref_torch as O
    from deflow_amd.synth import synth_batch
    torch.manual_seed(45)
    ref = O.DeFlow()
    mine = deflow_amd.DeFlow()
    mine.load_state_dict(ref.state_dict())
    mine = mine.to(dev).eval()
    ref.eval()
    batch = synth_batch(16, 80000, seed=777)
    bd = to_dev(batch, dev)
    with torch.no_grad():
        got32 = mine(bd)
        mine.inference_dtype = "bf16"
        got16 = mine(bd)
        worst = 0.0
        for b in range(16):
            emax, _ = _bf16_flow_err(got16["flow"][b], got32["flow"][b].cpu())
            worst = max(worst, emax)
        print(f"[parity] bf16 vs fp32 HIP forward, B=16 @512x512: worst max err / max|flow| = {worst:.2e} (tol 2e-2)")
        assert worst <= 2e-2
        for b in (0, 15):
            one = {k: v[b:b + 1] for k, v in batch.items()}
            want = ref(one)
            check(f"B=16 fp32 flow[{b}] vs oracle", got32["flow"][b], want["flow"][0], 1e-4)
            emax, erms = _bf16_flow_err(got16["flow"][b], want["flow"][0])
            print(f"[parity] bf16 B=16 flow[{b}] vs oracle: max {emax:.2e} rms {erms:.2e}")
            assert emax <= 2e-2


def test_bf16_training_trajectory(dev):
    """Trainer(dtype="bf16") -- BASELINE configs[4]'s "bf16 MFMA" as a TRAINING mode: UNet convolutions (forward, data and
    weight gradients) on bf16 MFMA operands with fp32 accumulation, fp32 master weights / Adam / BatchNorm statistics /
    decoder.  Six optimizer steps against the fp32 CPU oracle stepping torch.optim.Adam on the same batches: the loss
    trajectory must follow within 2e-2 relative (bf16 keeps 8 mantissa bits; measured ~1e-3) and must go down."""
    from oracle import ref_torch as O
    from deflow_amd.optim import Trainer
    ref, mine = build_pair(dev, 31, decoder_option="gru", num_iters=4)
    ref.train(); mine.train()
    opt = torch.optim.Adam(ref.parameters(), lr=2e-4)   # the README's learning rate (at 2e-3 the fp32 oracle itself diverges
    tr = Trainer(mine, lr=2e-4, dtype="bf16")            # within six steps and any rounding difference is amplified)
    batches = [make_batch(2, 1500, 5000 + 10 * i) for i in range(2)]
    want, got = [], []
    for i in range(6):
        b = batches[i % 2]
        opt.zero_grad()
        l = O.training_loss(ref(b), b)
        l.backward()
        opt.step()
        want.append(float(l.detach()))
        got.append(float(tr.step(to_dev(b, dev))))
    print("[parity] bf16 training loss trajectory:", [f"{g:.4f}/{w:.4f}" for g, w in zip(got, want)])
    import parity
    for i, (g, w) in enumerate(zip(got, want)):
        e = abs(g - w) / abs(w)
        parity.record("bf16_train", f"loss step {i}", err_vs_fp32_oracle=e, bound=2e-2, ok=e <= 2e-2)
        assert e <= 2e-2, (i, g, w)
    assert got[4] < got[0] and got[5] < got[1], "the loss on each of the two batches must go down"
    from deflow_amd import ops
    assert ops.MFMA_BF16 is False, "the switch must not leak out of Trainer.step"


def test_configs4_shape_training_step(dev, monkeypatch, golden_dir):
    """BASELINE configs[4] as a TRAINING shape, per GPU exactly as bench.py's `configs4_shape` runs it: 4 pairs, 1024 x 1024 grid
    (voxel 0.1 m), 160 000 points per cloud, 8 GRU iterations.  Against the ORACLE only: the committed digests of
    the fp32 and float64 oracle (oracle/gen_digest_bs16.py 1024 160000 --batch 4 --voxel 0.1 --iters 8; 83 GB float64 tape spilled to
    disk, ~45 CPU-minutes in the build container) -- fp32 step: every parameter gradient / flow / loss at max(1e-4, 4 x oracle-fp32)
    in the projection form of _bs16_case; then the bf16 training mode (bf16 MFMA operands + bf16 storage) against the same FLOAT64
    digests at the stated bound BF16_GRAD_RMS (round 3 compared it with the HIP fp32 step: a self-comparison)."""
    _bs16_case(dev, monkeypatch, 1024, 160000, "cfg4_train", golden_dir, digest="bs4_1024_it8_digest.npz", bf16=True)


def test_autograd_callers_keep_one_transpose_and_plane_set_per_weight(dev):
    """plain-autograd training (no Trainer): the transposed weights and the fp16 planes of every conv layer are kept per weight tensor
    while it is unchanged and REPLACED when an optimizer writes it (a backward made ~30 fresh transposes that never hit
    the cache and pinned a triple each until a 128-entry clear).  Three steps with torch.optim.SGD: the caches stop growing after
    the first step, a repeated backward on unchanged weights launches no transpose / split, and the gradients equal those of a run
    with the caches emptied before every step, bit for bit."""
    import copy
    from deflow_amd import ops
    _, base = build_pair(dev, 31, decoder_option="gru", num_iters=2)
    batch = to_dev(make_batch(2, 1500, 4100), dev)

    def run(clear):
        m = copy.deepcopy(base).train()
        opt = torch.optim.SGD(m.parameters(), lr=1e-3)
        sizes, grads = [], []
        for _ in range(3):
            if clear:
                ops._WT_CACHE.clear(); ops._PLANE_CACHE.clear()
            opt.zero_grad()
            res = m(batch)
            loss = sum((f ** 2).sum() for f in res["flow"])
            loss.backward()
            grads.append(torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone())
            sizes.append((len(ops._WT_CACHE), len(ops._PLANE_CACHE)))
            opt.step()
        return sizes, grads, m

    ops._WT_CACHE.clear(); ops._PLANE_CACHE.clear()
    sizes, g_cached, m = run(False)
    assert sizes[0][0] > 10 and sizes[1] == sizes[0] and sizes[2] == sizes[0], sizes        # one entry per weight tensor, replaced in place
    # unchanged weights: a second forward + backward finds every transpose and plane set
    names = []
    real = ops.call
    ops.call = lambda name, *a: (names.append(name), real(name, *a))[1]
    try:
        res = m(batch)
        sum((f ** 2).sum() for f in res["flow"]).backward()
        first = [n for n in names if n in ("df_weight_transpose", "df_split_h2")]
        names.clear()
        res = m(batch)
        sum((f ** 2).sum() for f in res["flow"]).backward()
        again = [n for n in names if n in ("df_weight_transpose", "df_split_h2")]
    finally:
        ops.call = real
    # (what remains: the GRU decoder's three packed gate matrices, built and transposed per call)
    assert first and "df_split_h2" not in again and len(again) <= 3, (len(first), again[:6])
    _, g_clear, _ = run(True)
    for a, b in zip(g_cached, g_clear):
        assert torch.equal(a, b)


BF16_GRAD_RMS_CONDITIONED = 5e-2     # bf16 training mode on CONDITIONED weights: per-tensor rms-relative error bound of every parameter gradient (round 5: worst projection 0.177 ||g|| = 3.9e-2 x 4.5, worst norm error 4.3e-2, both on the GRU head; median tensor 1.0e-2 / 1.6e-2 -- the 3e-2 VERDICT r4 asked for does not hold for the head's small tensors)


def test_bf16_gradients_on_conditioned_weights(dev, golden_dir):
    """the bf16 training mode's gradients against FLOAT64 at weights a run would have, not at the initialisation
    (where test_configs4_shape_training_step states 1e-1): 50 fp32 Adam steps of the HIP trainer (tests/helpers/conditioned_weights.py;
    loss 6.37 -> 4.5, BatchNorm running statistics moved), then ONE step at BASELINE configs[4]'s per-GPU shape (4 pairs, 1024 x 1024,
    160 000 points, 8 GRU iterations) in bf16 mode (bf16 MFMA operands + bf16 storage) against the float64 oracle's digest for those
    weights (oracle/gen_digest_bs16.py --weights, generated once in the build container from the GPU box's dump of the same 50 steps:
    the engine is deterministic).  Per tensor: 16 random-sign projections of the error, each N(0, ||e||^2):
        |proj(bf16) - proj(fp64)| <= 4.5 x 5e-2 x ||g_fp64||      and the gradient norm within 5e-2
    (BF16_GRAD_RMS_CONDITIONED; measured: the UNet's tensors 0.5-2e-2, the GRU head's gates / biases / offset encoder 3-4e-2 -- eight
    GRU iterations of bf16 operands on 128-wide rows; the 3e-2 the verdict hoped for holds for 81 of the 89 tensors, not for those);
    the fp32 step on the same weights is checked against the same digest at 1e-4 first (the weights ARE the digest's)."""
    import deflow_amd
    import parity
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers"))
    from conditioned_weights import conditioned_state
    from oracle.gen_digest_bs16 import project
    from deflow_amd.synth import synth_batch
    from deflow_amd.optim import Trainer
    from deflow_amd import ops
    dg = dict(np.load(os.path.join(golden_dir, "bs4_1024_it8_w50_digest.npz")))
    grid, n_pts, B, voxel, iters = int(dg["grid"]), int(dg["n_pts"]), int(dg["batch"]), float(dg["voxel"]), int(dg["iters"])
    assert (grid, n_pts, B, iters) == (1024, 160000, 4, 8)
    sd, losses = conditioned_state(dev)
    assert losses[-1] < 0.8 * losses[0], losses[::7]
    half = 0.5 * voxel * grid
    cfg = dict(voxel_size=[voxel, voxel, 6], point_cloud_range=[-half, -half, -3, half, half, 3], grid_feature_size=[grid, grid], num_iters=iters)
    bd = to_dev(synth_batch(B, n_pts, seed=int(dg["seed"]), grid_hw=(int(round(grid * voxel / 0.2)),) * 2, exact=True), dev)
    loss64 = float(dg["loss64"])

    def step(dtype):
        m = deflow_amd.DeFlow(**cfg)
        m.load_state_dict(sd)
        m = m.to(dev).train()
        tr = Trainer(m, lr=0.0, dtype=dtype)
        with ops.mfma_bf16(dtype == "bf16", dtype == "bf16" and tr.bf16_store), torch.no_grad():
            tr.flat.zero_grad(); tr.sink.begin()
            loss = tr._forward_backward(bd)
        torch.cuda.synchronize()
        out = {}
        for k, p in m.named_parameters():
            if parity.is_bn_shadowed_bias(k):
                continue
            l2 = float(dg[f"grad.{k}.l2"])
            dp = np.abs(project("grad." + k, p.grad).numpy() - dg[f"grad.{k}.proj"]).max()
            out[k] = (dp / max(l2, 1e-300), abs(float(p.grad.double().norm()) - l2) / max(l2, 1e-300))
        return float(loss), out

    l32, e32 = step("fp32")
    assert abs(l32 - loss64) <= 1e-4 * abs(loss64), (l32, loss64)
    w32 = max(e32.items(), key=lambda kv: kv[1][0])
    assert w32[1][0] <= 4.5 * 1e-4, ("the regenerated weights are not the digest's", w32)
    l16, e16 = step("bf16")
    worst = max(e16.items(), key=lambda kv: kv[1][0])
    worst_n = max(e16.items(), key=lambda kv: kv[1][1])
    for k, (dp, dn) in e16.items():
        parity.record("cfg4_w50_bf16", "grad " + k, max_proj_err_over_l2=dp, norm_err=dn, rms_bound=BF16_GRAD_RMS_CONDITIONED,
                      ok=dp <= 4.5 * BF16_GRAD_RMS_CONDITIONED and dn <= BF16_GRAD_RMS_CONDITIONED)
    print(f"[parity] conditioned weights (50 fp32 steps), configs[4] shape: fp32 worst projection error / ||g|| {w32[1][0]:.2e} ({w32[0]}); "
          f"bf16 loss {l16:.5f} / {loss64:.5f}, worst projection error / ||g|| {worst[1][0]:.3e} ({worst[0]}), worst norm error "
          f"{worst_n[1][1]:.3e} ({worst_n[0]}) -- bound 4.5 x {BF16_GRAD_RMS_CONDITIONED} / {BF16_GRAD_RMS_CONDITIONED}")
    assert abs(l16 - loss64) <= 2e-3 * abs(loss64), (l16, loss64)
    bad = [(k, v) for k, v in e16.items() if not (v[0] <= 4.5 * BF16_GRAD_RMS_CONDITIONED and v[1] <= BF16_GRAD_RMS_CONDITIONED)]
    assert not bad, bad[:6]


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_captured_step_equals_eager(dev, dtype):
    """Trainer.capture / step_captured: the whole training step (forward, loss, hand-sequenced backward, Adam -- several
    hundred launches) replayed as ONE HIP graph on new batches must leave the parameters, the Adam moments and the losses an
    eager Trainer leaves (the engine is deterministic; Adam's step number lives on the device in the captured form)."""
    import deflow_amd
    from deflow_amd.optim import Trainer
    from deflow_amd.synth import synth_batch

    def fresh():
        torch.manual_seed(77)
        m = deflow_amd.DeFlow(**SMALL, num_iters=2).to(dev).train()
        return m, Trainer(m, lr=1e-3, dtype=dtype)

    bs = [synth_batch(2, 1500, seed=300 + i, grid_hw=(64, 64), device=dev) for i in range(2)]
    seq = [bs[1], bs[0], bs[1], bs[1]]
    m1, t1 = fresh()
    want = [float(t1.step(b)) for b in seq]
    m2, t2 = fresh()
    t2.capture(bs[0])                       # its two warm-up launches leave no trace: the first replay is step 1
    assert t2.opt.step_count == 0 and [o[0] for o in t2._program] == ["graph"]      # one rank: ONE graph
    got = [float(t2.step_captured(b)) for b in seq]
    torch.cuda.synchronize()
    assert t2.opt.step_count == t1.opt.step_count == 4 and int(t2.opt.step_dev) == 4
    print(f"[parity] captured vs eager losses ({dtype}):", [f"{g:.6f}/{w:.6f}" for g, w in zip(got, want)])
    assert got == want, (got, want)
    check("captured params", t2.flat.param, t1.flat.param, 1e-6)
    check("captured exp_avg_sq", t2.opt.exp_avg_sq, t1.opt.exp_avg_sq, 1e-6)
    for (k, a), (_, b) in zip(m2.named_buffers(), m1.named_buffers()):
        assert torch.equal(a, b), k          # BatchNorm running statistics and counters moved inside the graph too
    with pytest.raises(ValueError):
        t2.step_captured(synth_batch(2, 1400, seed=1, grid_hw=(64, 64), device=dev))


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_direct_step_equals_autograd_step(dev, monkeypatch, dtype):
    """Trainer.step drives the engine directly (forward tape -> loss kernels -> autograd.deflow_backward on the caller's
    thread); DF_TRAINER_AUTOGRAD=1 takes the torch.autograd route (DeFlowFn + loss.backward(), what every other caller of the
    module uses).  Same launches in the same order: losses, parameters and Adam state must be bit-identical."""
    import deflow_amd
    from deflow_amd.optim import Trainer
    from deflow_amd.synth import synth_batch
    out = []
    for env in (None, "1"):
        if env:
            monkeypatch.setenv("DF_TRAINER_AUTOGRAD", env)
        torch.manual_seed(78)
        m = deflow_amd.DeFlow(**SMALL, num_iters=2).to(dev).train()
        t = Trainer(m, lr=1e-3, dtype=dtype)
        losses = [float(t.step(synth_batch(2, 1500, seed=310 + i, grid_hw=(64, 64), device=dev))) for i in range(3)]
        out.append((losses, t.flat.param.clone(), t.opt.exp_avg_sq.clone()))
    assert out[0][0] == out[1][0], (out[0][0], out[1][0])
    assert torch.equal(out[0][1], out[1][1]) and torch.equal(out[0][2], out[1][2])


@pytest.mark.parametrize("mode", ["graph", "graph_bf16"])
def test_captured_data_parallel_step_two_ranks(dev, tmp_path, mode):
    """the data-parallel step must be host-free.  Two ranks of the real engine on this GPU (gloo carries the
    collectives): Trainer.capture splits the step's HIP graph at the gradient buckets -- head, UNet decoder, encoder stages
    3 / 2 / 1, pillar net -- and replays [segment, all-reduce, segment, ...]; after three replays on changing batches every
    rank holds bit-for-bit what the eager data-parallel trainer holds, and a replay costs the host a few graph launches."""
    import socket
    import subprocess
    import sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "rank0.pt")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "tests", "helpers", "ddp_two_ranks_one_gpu.py"), out, mode],
                       cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    r0, r1 = torch.load(out)["ranks"]
    print(f"[ddp graph {mode}] program: {r0['kinds']}; host ms per replayed step (graph launches): {r0['host_ms']:.2f} / {r1['host_ms']:.2f}")
    for rr in (r0, r1):
        if mode == "graph":
            assert rr["got"] == rr["want"], (rr["got"], rr["want"])
            assert rr["same"], "captured data-parallel replay must leave the eager trainer's parameters / Adam state / buffers"
        else:
            # bf16 mode, TWO PROCESSES on one GPU: kernels with packed-fp32 instructions occasionally return wrong sums while the
            # other process runs bf16-MFMA kernels (a cross-process effect of this platform, tools/archive/pfn_bwd_stress.py; one
            # process per GPU -- the deployment -- is bit-reproducible, tools/archive/grad_repro_probe.py): bounded, not bit-exact, here
            assert all(abs(g - w) <= 1e-3 * abs(w) for g, w in zip(rr["got"], rr["want"])), (rr["got"], rr["want"])
            assert rr["diag"]["param"] <= 5e-3 and rr["diag"]["buffers"] <= 1e-3, rr["diag"]
        assert rr["steps"] == (3, 3, 3)
        assert rr["n_graph"] >= 6 and rr["n_allreduce"] >= 6 and rr["kinds"][-2:] == ["wait", "graph"]
        # (host time of a replayed step: ~1-1.5 ms on an idle box; the bound is a sanity check that replay does not re-enqueue the ~330
        #  launches of an eager step -- tens of ms of host time -- and is kept wide: a loaded host of the pool measured 5.3 ms)
        assert rr["host_ms"] < 20.0
    assert r0["param_sum"] == r1["param_sum"]


def test_train_mode_forward_without_grad_is_repeatable():
    """model.train() under torch.no_grad() keeps no tape: layer outputs must still outlive the kernels that read them
    (regression: the UNet freed each activation as soon as the next layer's buffers were allocated, and the allocator
    handed the block to the next conv's output -> run-to-run differences of a few percent)"""
    import deflow_amd
    dev = torch.device("cuda")
    torch.manual_seed(3)
    m = deflow_amd.DeFlow(grid_feature_size=[256, 256], point_cloud_range=[-25.6, -25.6, -3, 25.6, 25.6, 3]).to(dev).train()
    from deflow_amd.synth import synth_batch
    batch = synth_batch(2, 20000, grid_hw=(256, 256), device=dev)
    with torch.no_grad():
        outs = [m.forward_padded(batch)["flow"].clone() for _ in range(3)]
    n = int(m.last_state["counts0"][0])
    assert n > 1000
    assert torch.equal(outs[0][0, :n], outs[1][0, :n]) and torch.equal(outs[0][0, :n], outs[2][0, :n])


def test_train_step_gradients_bit_reproducible_under_memory_poison():
    """the gradient arena of a full training step is bit-identical when the step is repeated, also after every free
    block of the caching allocator was filled with NaN (no kernel may read memory it did not write: torch.empty
    workspaces, partial-sum buffers, saved GRU planes of invalid rows, DMA zero-fill regions ...)"""
    import deflow_amd
    from deflow_amd.optim import Trainer
    from deflow_amd.synth import synth_batch
    dev = torch.device("cuda")
    torch.manual_seed(5)
    m = deflow_amd.DeFlow(grid_feature_size=[256, 256], point_cloud_range=[-25.6, -25.6, -3, 25.6, 25.6, 3]).to(dev).train()
    tr = Trainer(m, lr=2e-4)
    batch = synth_batch(2, 20000, grid_hw=(256, 256), device=dev)

    def grads():
        tr.flat.zero_grad()
        tr.sink.begin()
        m.forward_padded(batch)
        loss = tr.loss_on_last_forward(batch)
        loss.backward()
        return tr.flat.grad.clone(), float(loss.detach())

    g0, l0 = grads()
    g1, l1 = grads()
    m.last_state = None
    torch.cuda.empty_cache()
    junk = torch.full((1024 ** 3,), float("nan"), device=dev)  # 4 GB of NaN back into the allocator's free lists
    del junk
    g2, l2 = grads()
    assert l0 == l1 == l2 and np.isfinite(l0)
    assert torch.equal(g0, g1), "repeated step must be bit-identical (no atomics, fixed reduction orders)"
    assert torch.equal(g0, g2) and not torch.isnan(g2).any(), "a kernel read memory it never wrote"
    assert float(g0.abs().max()) > 0


@pytest.mark.parametrize("B,N,grid", [(1, 1000, 64), (3, 4097, 128), (5, 777, 64), (2, 20000, 256)])
def test_sparse_edge_kernels_match_dense(B, N, grid):
    """the four sparse kernels at the UNet's ends (df_pillar_input_grad, df_sparse_in_wgrad, df_sparse_conv3x3,
    df_sparse_wgrad3x3) against the dense kernels they replace: whole gradient arena of a training step, on odd batch
    sizes / point counts, with an all-NaN sample and a 5-point sample"""
    import deflow_amd
    from deflow_amd.optim import Trainer
    from deflow_amd.synth import synth_batch
    dev = torch.device("cuda")
    torch.manual_seed(B * 7 + N)
    rng = [-0.1 * grid, -0.1 * grid, -3, 0.1 * grid, 0.1 * grid, 3]
    m = deflow_amd.DeFlow(grid_feature_size=[grid, grid], point_cloud_range=rng).to(dev).train()
    tr = Trainer(m, lr=2e-4)
    batch = synth_batch(B, N, grid_hw=(grid, grid), device=dev)
    if B >= 3:
        batch["pc0"][1] = float("nan")
        batch["pc0"][2, 5:] = float("nan")
    out = {}
    old = os.environ.get("DF_DENSE_CANVAS_GRAD")
    try:
        for mode in ("0", "1"):
            os.environ["DF_DENSE_CANVAS_GRAD"] = mode
            tr.flat.zero_grad()
            tr.sink.begin()
            m.forward_padded(batch)
            loss = tr.loss_on_last_forward(batch)
            loss.backward()
            out[mode] = (tr.flat.grad.clone(), float(loss.detach()))
    finally:
        if old is None:
            os.environ.pop("DF_DENSE_CANVAS_GRAD", None)
        else:
            os.environ["DF_DENSE_CANVAS_GRAD"] = old
    g0, g1 = out["0"][0], out["1"][0]
    assert torch.isfinite(g0).all() and out["0"][1] == pytest.approx(out["1"][1], rel=1e-6)
    err = float((g0 - g1).abs().max() / g1.abs().max())
    print(f"[parity] sparse vs dense edge kernels B={B} N={N} grid={grid}: {err:.2e}")
    assert err < 1e-5


# switches of DIFFERENT kernel families are combined in one subprocess (they do not interact); switches that select between
# forms of the SAME kernel get their own run.  Four runs of ~30 s instead of seven.
_X3_OFF = {"DF_CONV_X3": "0", "DF_WGRAD_X3": "0"}      # the fp32-MFMA kernels the bf16x3 forms replaced by default (round 3)


@pytest.mark.slow
@pytest.mark.parametrize("env", [{"DF_CONV_NO_DMA": "1", "DF_CONV_WIDE_EPI": "1", "DF_GRU_V1": "1", "DF_GRU_WGRAD_V1": "1", **_X3_OFF},
                                 {"DF_WGRAD_DMA_ALL": "1", "DF_CONV_HALO": "0", "DF_DENSE_CANVAS_GRAD": "1", "DF_GRU_X2": "0", **_X3_OFF},
                                 {"DF_WGRAD_RING": "0", "DF_WGRAD_RING_S2": "0", "DF_CONV_W8": "0", "DF_CONV_HALO": "0", "DF_SIDE_STREAM": "1", **_X3_OFF},
                                 {"DF_WGRAD_RING": "3", "DF_MERGE_CLOUDS": "0", "DF_NO_FUSED_BIAS": "1", **_X3_OFF},
                                 {"DF_CONV_H2": "0", "DF_CONV_X3_BM256": "0"}])      # the bf16x3 forms the fp16x2 ones replaced by default
def test_alternate_kernel_paths(env):
    """the register-staged conv/wgrad kernels (fallback for > 4 GB tensors), the all-DMA wgrad variants, the
    first-generation GRU kernels with unfused gate weight gradients, and the side-stream weight-gradient schedule stay
    correct: re-run the conv / ConvWithNorms / decoder parity tests and THE train-step parity test (one model test per leg: each
    pays the CPU oracle in fp32 and float64 again) in a subprocess with the override (the
    bf16-operand tests are left out: the alternate kernel forms have no bf16 mode and run fp32 when it is requested)"""
    import subprocess
    import sys
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "tests/test_gpu_kernels.py", "tests/test_gpu_model.py",
                        "-k", "(conv or cwn or gru or test_train_step_vs_oracle) and not full_size and not bs16 and not bf16 and not x3 and not h2", "-p", "no:cacheprovider"], env=e, capture_output=True, text=True,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_rccl_collective_path_on_a_one_rank_group():
    """the data-parallel path over RCCL itself (not gloo): process group on the device, arena broadcast, the bucketed
    asynchronous all-reduces issued inside the backward, the wait before Adam -- on the 1-rank group a 1-GPU box allows,
    where the result must be bit-identical to a trainer that issues no collectives (tests/helpers/rccl_world1.py)."""
    import socket
    import subprocess
    import sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(root, "tests", "helpers", "rccl_world1.py")],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
    print(r.stdout[-2000:], r.stderr[-3000:])
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout and "RCCL_WORLD1_GRAPH_OK" in r.stdout and "RCCL_WORLD1_TRACE_OK" in r.stdout


def test_training_from_scene_files(dev, tmp_path, capsys):
    """section 8(f) N2 end to end: the trainer fed by the HDF5 scene fixtures (written by real h5py) through the in-tree
    reader, NaN-pad collate, sharded sampler and prefetching loader, incl. node-local staging; the batches the model sees
    equal a direct collate of the dataset items, and the loss stays finite over an epoch."""
    import json
    from deflow_amd import train as T
    from deflow_amd.data import HDF5Dataset, SceneLoader, ShardedSampler, collate_fn_pad
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "av2_mini", "train")
    ds = HDF5Dataset(root)
    sampler = ShardedSampler(len(ds), 0, 1, shuffle=True, seed=3)
    for k, b in enumerate(SceneLoader(ds, 4, sampler, device=dev, num_workers=2)):
        want = collate_fn_pad([ds[i] for i in list(sampler)[4 * k: 4 * k + 4]])
        assert b["pc0"].is_cuda and torch.equal(torch.nan_to_num(b["pc0"]).cpu(), torch.nan_to_num(want["pc0"]))
        assert torch.equal(torch.nan_to_num(b["flow"]).cpu(), torch.nan_to_num(want["flow"]))
        if k == 3:
            break
    T.main(["model=deflow", "lr=2e-4", "epochs=1", "batch_size=4", "loss_fn=deflowLoss", "model.target.num_iters=2",
            "voxel_size=[0.4, 0.4, 6]", f"train_data={root}", f"val_data={root}", "num_workers=1",
            f"stage_dir={tmp_path / 'scratch'}", "log_every=1", f"save_checkpoint={tmp_path / 'm.ckpt'}"])
    lines = [json.loads(l) for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    steps = [l for l in lines if "trainer/loss" in l]
    assert len(steps) == 95 // 4 and all(np.isfinite(l["trainer/loss"]) for l in steps)
    val = [l for l in lines if "val" in l]
    assert val and np.isfinite(val[-1]["val"]["EPE"])
    assert sorted(os.listdir(tmp_path / "scratch" / "train")) == sorted(os.listdir(root))
    # the reference's evaluation entry: checkpoint + data, configuration restored from the checkpoint
    from deflow_amd import eval as E
    m = E.main([f"checkpoint={tmp_path / 'm.ckpt'}", "av2_mode=val", f"val_data={root}", "num_workers=0"])
    assert np.isfinite(m["EPE"]) and abs(m["EPE"] - val[-1]["val"]["EPE"]) < 5e-2 and m["n"] > 0


@pytest.mark.parametrize("loss_fn", ["ff3dLoss", "zeroflowLoss"])
@pytest.mark.parametrize("path", ["autograd", "kernels"])
def test_ablation_losses_vs_oracle(dev, loss_fn, path):
    """loss_fn=ff3dLoss / zeroflowLoss ([REF 1_train.sh:58-78]) on a labelled batch of the scene fixtures: the trainer's
    loss on the padded device tensors and the gradients it sends back through the engine vs the oracle's per-sample form."""
    from oracle import ref_torch as O
    from deflow_amd.data import HDF5Dataset, collate_fn_pad
    from deflow_amd.optim import Trainer
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "av2_mini", "train")
    ds = HDF5Dataset(root)
    batch = collate_fn_pad([ds[3], ds[40], ds[94], ds[70]])          # item 94 has an empty pc0
    import parity
    ref, mine = build_pair(dev, 13, decoder_option="gru", num_iters=2)
    ref.train(); mine.train()
    ref, ref64 = parity.oracle_pair(ref)
    o32, o64 = parity.oracle_step(ref, batch, loss_fn), parity.oracle_step(ref64, batch, loss_fn)
    tr = Trainer(mine, lr=2e-4, loss_fn=loss_fn)
    bd = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    tr.flat.zero_grad(); tr.sink.begin()
    if path == "autograd":      # the torch form of deflow_amd/losses.py through the autograd engine
        mine.forward_padded(bd)
        loss_m = tr.loss_on_last_forward(bd)
        loss_m.backward()
    else:                       # the Trainer's direct step: df_wloss_fwd / _finalize / _bwd (round 5)
        names = []
        from deflow_amd import optim as _optim
        real = _optim.call
        _optim.call = lambda name, *a: (names.append(name), real(name, *a))[1]
        try:
            with torch.no_grad():
                loss_m = tr._forward_backward(bd)
        finally:
            _optim.call = real
        assert {"df_wloss_fwd", "df_wloss_finalize", "df_wloss_bwd"} <= set(names), [n for n in names if "loss" in n]
    st = mine.last_state
    m0 = st["counts0"].tolist()
    res_m = {"flow": [st["flow"][b, :m0[b]] for b in range(len(m0))]}
    parity.check_step(f"ablation_{loss_fn}_{path}", mine, res_m, loss_m.detach(), o32, o64)   # fp64 three-way bound, every gradient


@pytest.mark.parametrize("train", [False, True])
def test_merged_cloud_pillarisation_is_bit_identical(dev, train):
    """tape-less forwards pillarise pc0 and pc1 as one set of 2B samples (half the launches): flows, valid indices and --
    in train mode under no_grad -- the BatchNorm running statistics must equal the two-call form bit for bit"""
    import copy
    _, base = build_pair(dev, 21, decoder_option="gru", num_iters=2)
    batch = to_dev(make_batch(3, 2000, 900), dev)
    batch["pc0"][1, 700:] = float("nan")                      # ragged validity
    out = {}
    for mode in ("0", "1"):
        m = copy.deepcopy(base)
        m.train(train)
        os.environ["DF_MERGE_CLOUDS"] = mode
        try:
            with torch.no_grad():
                st = m.forward_padded(batch)
                res = m(batch)
        finally:
            os.environ.pop("DF_MERGE_CLOUDS", None)
        out[mode] = (st["flow"].clone(), st["counts0"].clone(), st["counts1"].clone(), res,
                     {k: v.clone() for k, v in m.state_dict().items() if "running" in k or "num_batches" in k})
    a, b = out["0"], out["1"]
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    for s in range(3):
        n = int(a[1][s])
        assert torch.equal(a[0][s, :n], b[0][s, :n])
        for key in ("flow", "pc0_valid_point_idxes", "pc1_valid_point_idxes", "pc1_points_lst", "pc0_points_lst"):
            assert torch.equal(a[3][key][s], b[3][key][s]), key
    for k in a[4]:
        assert torch.equal(a[4][k], b[4][k]), k


def _varied_batches(dev):
    """batches whose occupied cells differ a lot from one to the next (sparser, denser, shifted, ragged, an empty cloud)"""
    out = []
    for i, (n, seed) in enumerate([(2000, 900), (600, 77), (2600, 5), (2000, 1234), (2000, 900)]):
        b = make_batch(3, n, seed)
        if i == 1:
            b["pc0"][0] = float("nan")                  # an empty cloud: every cell it occupied before must be zero again
            b["pc1"][2, 100:] = float("nan")
        if i == 2:
            b["pc0"][..., 0] += 3.1                       # shifted: other cells
        if i == 3:
            b["pc1"][1, 900:] = float("nan")
        out.append(to_dev(b, dev))
    return out


@pytest.mark.parametrize("train", [False, True])
def test_persistent_canvas_equals_the_dense_one(dev, train, monkeypatch):
    """round 5: no-grad forwards keep ONE BEV canvas per (model, shape) and the band kernel rewrites only the cells occupied now or
    last time (df_pillar2_band_sp).  Over a sequence of different batches: the canvas equals a freshly, densely written one bit for
    bit, and so do the flows; a batch-size change in between (another canvas) does not disturb the first one's invariant."""
    import copy
    from deflow_amd import deflow as D
    from deflow_amd._lib import img
    _, base = build_pair(dev, 22, decoder_option="gru", num_iters=2)
    batches = _varied_batches(dev)
    small = to_dev(make_batch(1, 1500, 31), dev)
    flows = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("DF_CANVAS_PERSIST", mode)
        m = copy.deepcopy(base).train(train)
        res = []
        with torch.no_grad():
            for i, b in enumerate(batches):
                st = m.forward_padded(b)
                res.append((st["flow"].clone(), st["counts0"].clone()))
                if i == 2:
                    m.forward_padded(small)
                if mode == "1":
                    # the persistent canvas itself against a dense write of the same clouds
                    store = D._CANVASES[m]
                    (canvas, _), = [v for k, v in store.items() if k[0] == 3]
                    emb = m.embedder
                    dense = torch.full_like(canvas, float("nan"))
                    keep = [t.clone() for t in emb.buffers()]       # (train mode: the dense reference must not move the running statistics)
                    emb.pillarize(st["pc0s"], img(dense, 32, 0), train, need_cells=False)      # (pc0 after the ego-motion transform)
                    emb.pillarize(b["pc1"], img(dense, 32, 32), train, need_cells=False)
                    for t, v in zip(emb.buffers(), keep):
                        t.copy_(v)
                    assert torch.equal(canvas, dense), (i, int((canvas != dense).sum()))
        flows[mode] = res
        assert (m in D._CANVASES) == (mode == "1")
    for i, (a, b) in enumerate(zip(flows["1"], flows["0"])):
        assert torch.equal(a[1], b[1]), i
        for s in range(3):
            n = int(a[1][s])
            assert torch.equal(a[0][s, :n], b[0][s, :n]), (i, s)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_trainer_persistent_canvas_steps_equal_dense_steps(dev, dtype, monkeypatch):
    """the trainer's step keeps its canvas across steps (one forward, then its backward): five steps on five different batches
    leave bit-identical parameters and losses with DF_CANVAS_PERSIST=1 and =0; an evaluation forward between two steps (its own
    canvas: the merged 2B-sample form) changes nothing"""
    import copy
    from deflow_amd.optim import Trainer
    _, base = build_pair(dev, 23, decoder_option="gru", num_iters=2)
    batches = _varied_batches(dev)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("DF_CANVAS_PERSIST", mode)
        m = copy.deepcopy(base).train()
        tr = Trainer(m, lr=1e-3, dtype=dtype)
        losses = []
        for i, b in enumerate(batches):
            losses.append(tr.step(b).clone())
            if i == 1:
                m.eval()
                with torch.no_grad():
                    m.forward_padded(batches[3])
                m.train()
        out[mode] = (torch.stack(losses), tr.flat.param.clone(), tr.flat.grad.clone())
    for a, b in zip(out["1"], out["0"]):
        assert torch.equal(a, b), float((a - b).abs().max())


def test_config0_fastflow3d_ff3dloss_bs1_from_scene_files(dev, tmp_path, capsys):
    """BASELINE configs[0] ("fastflow3d model, ... single AV2 scene, batch_size=1", the README's baseline command
    [REF README.md:68]) as plumbing through this engine: model=fastflow3d (LinearDecoder head) with loss_fn=ff3dLoss at
    batch size 1 over one scene file, then the evaluation entry on the checkpoint it wrote.  (num_workers=0 throughout: forking
    loader workers out of a pytest process that has run 150 GPU tests costs seconds per fork -- 70 of this test's 86 s in round 3;
    the worker-process path is test_training_from_scene_files)"""
    import json
    import pickle
    import shutil
    from deflow_amd import train as T, eval as E
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "av2_mini", "train")
    one = tmp_path / "one_scene"
    one.mkdir()
    shutil.copy(os.path.join(root, "scene_a.h5"), one / "scene_a.h5")
    index = [e for e in pickle.load(open(os.path.join(root, "index_total.pkl"), "rb")) if e[0] == "scene_a"]
    pickle.dump(index, open(one / "index_total.pkl", "wb"))
    # the reference's data root layout [REF 1_train.sh:12-14; 2_eval.sh:13]: <dataset_path>/train and <dataset_path>/val
    sensor = tmp_path / "av2" / "sensor"
    shutil.copytree(one, sensor / "train")
    shutil.copytree(one, sensor / "val")
    ck = tmp_path / "ff3d.ckpt"
    T.main(["model=fastflow3d", "lr=4e-5", "epochs=1", "batch_size=1", "loss_fn=ff3dLoss", "voxel_size=[0.4, 0.4, 6]",
            f"dataset_path={sensor}", "num_workers=0", "log_every=4", f"save_checkpoint={ck}"])
    lines = [json.loads(l) for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    steps = [l for l in lines if "trainer/loss" in l]
    assert len(steps) == len(index) // 4 and all(np.isfinite(l["trainer/loss"]) for l in steps)
    val = [l for l in lines if "val" in l]
    m = E.main([f"checkpoint={ck}", "av2_mode=val", f"val_data={one}", "num_workers=0"])
    assert np.isfinite(m["EPE"]) and m["n"] > 0
    # the reference's LITERAL evaluation command [REF assets/slurm/2_eval.sh:33-35]:
    #   eval.py wandb_mode=online dataset_path=/scratch/local/av2/sensor av2_mode=val checkpoint=<ckpt>
    # must read <dataset_path>/val (round 2 silently evaluated synthetic pairs here)
    m2 = E.main(["wandb_mode=online", f"dataset_path={sensor}", "av2_mode=val", f"checkpoint={ck}", "num_workers=0"])
    line = json.loads([l for l in capsys.readouterr().out.splitlines() if l.startswith("{")][-1])
    assert line["val_data"] == str(sensor / "val") and line["model"] == "fastflow3d"
    assert m2["n"] == m["n"] and abs(m2["EPE"] - m["EPE"]) < 1e-6 and abs(m2["EPE"] - val[-1]["val"]["EPE"]) < 1e-3
    with pytest.raises(Exception):      # a data key that leads nowhere must not fall back to synthetic pairs
        E.main([f"dataset_path={tmp_path / 'nowhere'}", "av2_mode=val", f"checkpoint={ck}"])
    # a checkpoint whose configuration is NESTED the way the reference's hydra / Lightning files are (model.name,
    # model.target.*): the architecture must come back from it (was built as DeFlow and loaded with strict=False)
    sd = torch.load(ck, map_location="cpu", weights_only=False)
    sd["hyper_parameters"] = {"cfg": {"model": {"name": "fastflow3d", "target": {"_target_": "scripts.network.models.fastflow3d.FastFlow3D",
                                                                                "voxel_size": [0.4, 0.4, 6], "num_iters": 4}},
                                      "voxel_size": [0.4, 0.4, 6], "point_cloud_range": [-51.2, -51.2, -3, 51.2, 51.2, 3],
                                      "batch_size": 1, "lr": 4e-5, "dataset_path": "/somewhere/else"}}
    nested = tmp_path / "nested.ckpt"
    torch.save(sd, nested)
    m3 = E.main([f"dataset_path={sensor}", "av2_mode=val", f"checkpoint={nested}", "num_workers=0"])
    cap = capsys.readouterr()
    assert "mismatch" not in cap.err and abs(m3["EPE"] - m["EPE"]) < 1e-6


@pytest.mark.parametrize("version", [1, 2])
def test_eval_leaderboard_tables_vs_oracle(dev, tmp_path, capsys, version):
    """Row N3: ``python -m deflow_amd.eval checkpoint=... av2_mode=val leaderboard_version=1|2`` [REF README.md:88-91;
    assets/slurm/2_eval.sh:33-35] on labelled scene files: every number of the leaderboard table (three-way EPE / IoU / accuracies /
    angle error inside the 35 m box; bucketed normalised EPE per meta-class) against oracle/ref_metrics.py evaluated on the ORACLE
    model's flow (oracle/ref_torch.py, same weights, same sweeps, CPU).  The flows agree to ~1e-5, so do the tables."""
    import json
    import shutil
    from deflow_amd import eval as E
    from deflow_amd.data import HDF5Dataset, collate_fn_pad
    from oracle import ref_metrics as R
    from oracle import ref_torch as O
    # tests/golden/av2_mini/val (gen_h5_val_fixture.py): labels consistent with the ego motion, every meta-class and speed bucket,
    # points on both sides of the 35 m range, unlabelled points, an eval_mask dataset
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "av2_mini", "val")
    val = tmp_path / "sensor" / "val"
    shutil.copytree(src, val)
    cfg = dict(voxel_size=[0.4, 0.4, 6], point_cloud_range=[-51.2, -51.2, -3, 51.2, 51.2, 3], grid_feature_size=[256, 256],
               decoder_option="gru", num_iters=2)
    torch.manual_seed(77)
    ref = O.DeFlow(**cfg).eval()
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                m.weight.uniform_(0.6, 1.4); m.bias.uniform_(-0.2, 0.2)
                m.running_mean.uniform_(-0.3, 0.3); m.running_var.uniform_(0.6, 1.5)
    ck = tmp_path / "model.ckpt"
    torch.save({"state_dict": {"model." + k: v for k, v in ref.state_dict().items()},
                "hyper_parameters": {"cfg": {"model": {"name": "deflow", "target": {"num_iters": 2, "decoder_option": "gru"}},
                                             "voxel_size": [0.4, 0.4, 6], "point_cloud_range": cfg["point_cloud_range"], "batch_size": 4}}}, ck)
    out = E.main([f"checkpoint={ck}", "av2_mode=val", f"dataset_path={tmp_path / 'sensor'}", "num_workers=0", "batch_size=4",
                  f"leaderboard_version={version}"])
    cap = capsys.readouterr()
    line = json.loads([l for l in cap.out.splitlines() if l.startswith("{")][-1])
    assert line["leaderboard_version"] == version and ("Three-way" if version == 1 else "mean/Dynamic") in line["leaderboard"]
    assert ("Three-way" if version == 1 else "WHEELED_VRU") in cap.err          # the printed table
    # the oracle: same sweeps through the CPU model, metrics by the numpy restatement
    ds = HDF5Dataset(str(val))
    om = R.OfficialMetrics()
    with torch.no_grad():
        for i0 in range(0, len(ds), 4):
            batch = collate_fn_pad([ds[i] for i in range(i0, min(i0 + 4, len(ds)))])
            res = ref(batch)
            for b in range(len(res["flow"])):
                vi = res["pc0_valid_point_idxes"][b]
                pf = res["pose_flow"][b][vi]
                a = [(pf + res["flow"][b]).double().numpy(), pf.double().numpy(), batch["pc0"][b][vi].double().numpy(),
                     batch["flow"][b][vi].double().numpy(), (batch["flow_is_valid"][b][vi] & batch["eval_mask"][b][vi]).numpy(),
                     batch["flow_category_indices"][b][vi].numpy()]
                om.step(R.evaluate_leaderboard(*a), R.evaluate_leaderboard_v2(*a))
    want, got = om.result(version), out["leaderboard"]
    assert set(want) == set(got)
    n_checked = 0
    for k, w in want.items():
        g = got[k]
        if isinstance(w, float) and np.isnan(w):
            assert np.isnan(g), k
            continue
        assert abs(g - w) <= 1e-3 * max(1.0, abs(w)), (k, g, w)       # flows differ by ~1e-5; thresholded counts by at most a point
        n_checked += 1
    # the fixture fills every cell of both tables except BACKGROUND/Dynamic (unlabelled points do not move)
    assert n_checked == len(want) - (1 if version == 2 else 0) and (version == 1 or np.isnan(want["BACKGROUND/Dynamic"]))
    if version == 1:
        assert got["n"] == want["n"] and got["n"] > 1000


def test_gradient_clipping_matches_clip_grad_norm(dev):
    """Trainer(gradient_clip_val=c) = torch.nn.utils.clip_grad_norm_(params, c) before Adam: parameters after two steps vs the
    oracle trained with torch's own clipping"""
    from oracle import ref_torch as O
    from deflow_amd.optim import Trainer
    ref, m = build_pair(dev, 31, decoder_option="gru", num_iters=2)
    ref.train(); m.train()
    tr = Trainer(m, lr=2e-4, gradient_clip_val=0.5)
    opt = torch.optim.Adam(ref.parameters(), lr=2e-4)
    batch_cpu = make_batch(2, 1500, 640)
    batch = to_dev(batch_cpu, dev)
    for _ in range(2):
        tr.step(batch)
        opt.zero_grad()
        O.training_loss(ref(batch_cpu), batch_cpu).backward()
        n = torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.5)
        assert float(n) > 0.5          # the clip is active in this test
        opt.step()
    pr = dict(ref.named_parameters())
    worst = max(rel_err(p_, pr[k]) for k, p_ in m.named_parameters() if not (k.endswith("conv.bias") and "encoder_step" in k))
    assert worst < 5e-3, worst


def test_two_data_parallel_ranks_on_one_gpu(dev, tmp_path):
    """the N > 1 training step with the real kernels: two ranks share the GPU (gloo carries the collectives), each with its
    own shard; parameters after two steps must equal the oracle stepping Adam on the AVERAGE of the two per-rank gradients
    (per-rank BatchNorm statistics), both ranks must end with identical parameters, and the bucketed asynchronous
    all-reduces must really have been issued from inside the backward"""
    import socket
    import subprocess
    import sys
    from oracle import ref_torch as O
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "rank0.pt")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "tests", "helpers", "ddp_two_ranks_one_gpu.py"), out],
                       cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    got = torch.load(out)
    r0, r1 = got["ranks"]
    assert r0["works"] >= 4 and r1["works"] >= 4 and r0["param_sum"] == r1["param_sum"]
    # oracle: rank 0's initial weights, gradients of the two shards averaged, Adam
    ref, _ = build_pair(dev, 41, decoder_option="gru", num_iters=2)
    ref.train()
    opt = torch.optim.Adam(ref.parameters(), lr=2e-4)
    shards = [make_batch(2, 1500, 7000 + 50 * k) for k in range(2)]
    want_losses = [[], []]
    for _ in range(2):
        opt.zero_grad()
        grads = None
        for k, b in enumerate(shards):
            for p in ref.parameters():
                p.grad = None
            l = O.training_loss(ref(b), b)
            l.backward()
            want_losses[k].append(float(l.detach()))
            g = [p.grad.clone() for p in ref.parameters()]
            grads = g if grads is None else [a + c for a, c in zip(grads, g)]
        for p, g in zip(ref.parameters(), grads):
            p.grad = g / 2
        opt.step()
    for k, rk in enumerate((r0, r1)):
        for a, w in zip(rk["losses"], want_losses[k]):
            assert abs(a - w) <= 2e-3 * abs(w), (k, rk["losses"], want_losses[k])
    pr = dict(ref.named_parameters())
    worst = max(rel_err(got["state"][k], p.detach()) for k, p in pr.items() if not (k.endswith("conv.bias") and "encoder_step" in k))
    print(f"[ddp] two ranks on one GPU: worst parameter rel diff after 2 steps {worst:.3e}")
    assert worst < 5e-3, worst


def _oracle_union_embedder(ref, n_ranks: int, b_local: int):
    """make the oracle's embedder compute what sync_bn computes across ranks on a rank-major union batch: the pillar feature
    net's BatchNorm1d is called once per sample, and call b of every rank shares its batch statistics -- i.e. the statistics
    run over the points of samples (rank 0, b), (rank 1, b), ... together"""
    from oracle import ref_torch as O
    emb, fn = ref.embedder, ref.embedder.feature_net

    def forward(points):
        infos = emb.voxelizer(points)
        feats = []
        for info in infos:
            pts, coors = info["points"], info["voxel_coords"]
            vmean, _, inv = O._scatter_mean(pts, coors)
            f_cluster = pts[:, :3] - vmean[inv][:, :3]
            f_center = pts.new_zeros((pts.shape[0], 3))
            f_center[:, 0] = pts[:, 0] - (coors[:, 2].type_as(pts) * fn.vx + fn.x_offset)
            f_center[:, 1] = pts[:, 1] - (coors[:, 1].type_as(pts) * fn.vy + fn.y_offset)
            f_center[:, 2] = pts[:, 2] - (coors[:, 0].type_as(pts) * fn.vz + fn.z_offset)
            feats.append(torch.cat([pts, f_cluster, f_center], dim=-1))
        outs = [None] * len(infos)
        for b in range(b_local):
            idx = [r * b_local + b for r in range(n_ranks)]
            y = fn.pfn_layers[0](torch.cat([feats[i] for i in idx]))
            off = 0
            for i in idx:
                outs[i] = y[off: off + feats[i].shape[0]]
                off += feats[i].shape[0]
        imgs = []
        for info, pf in zip(infos, outs):
            vf, vc, _ = O._scatter_mean(pf, info["voxel_coords"])
            imgs.append(emb.scatter(vf, vc))
        return torch.cat(imgs, dim=0), infos
    emb.forward = forward


def test_sync_bn_two_ranks_on_one_gpu(dev, tmp_path):
    """sync_bn=True over two ranks of the real engine (sharing the GPU, gloo collectives) = ONE model seeing the union batch:
    BatchNorm2d statistics over both shards per encoder call, the pillar net's BatchNorm1d over the points of sample b of both
    ranks; parameters after two steps vs the oracle on the union batch with the loss averaged over the ranks"""
    import socket
    import subprocess
    import sys
    from oracle import ref_torch as O
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "rank0.pt")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "tests", "helpers", "ddp_two_ranks_one_gpu.py"), out, "sync_bn"],
                       cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    got = torch.load(out)
    r0, r1 = got["ranks"]
    assert r0["param_sum"] == r1["param_sum"]
    ref, _ = build_pair(dev, 41, decoder_option="gru", num_iters=2)
    ref.train()
    _oracle_union_embedder(ref, 2, 2)
    opt = torch.optim.Adam(ref.parameters(), lr=2e-4)
    shards = [make_batch(2, 1500, 7000 + 50 * k) for k in range(2)]
    union = {k: torch.cat([b[k] for b in shards]) for k in shards[0]}
    want = []
    for _ in range(2):
        opt.zero_grad()
        res = ref(union)
        per_rank = [O.training_loss({k: v[2 * rk: 2 * rk + 2] for k, v in res.items()}, shards[rk]) for rk in range(2)]
        want.append([float(l.detach()) for l in per_rank])
        ((per_rank[0] + per_rank[1]) / 2).backward()
        opt.step()
    for step in range(2):
        for rk, rr in enumerate((r0, r1)):
            assert abs(rr["losses"][step] - want[step][rk]) <= 2e-3 * abs(want[step][rk]), (rr["losses"], want)
    pr = dict(ref.named_parameters())
    worst = max(rel_err(got["state"][k], p.detach()) for k, p in pr.items() if not (k.endswith("conv.bias") and "encoder_step" in k))
    print(f"[ddp] sync_bn, two ranks on one GPU: worst parameter rel diff after 2 steps {worst:.3e}")
    assert worst < 5e-3, worst
    # and it is NOT what per-rank statistics give: the first-step losses of the per-rank-BN run differ from these
    alone = O.training_loss(build_pair(dev, 41, decoder_option="gru", num_iters=2)[0].train()(shards[0]), shards[0])
    assert abs(want[0][0] - float(alone.detach())) > 1e-4


def test_cli_two_ranks_share_the_scene_files(dev, tmp_path):
    """`torch.distributed.run --nproc-per-node 2 -m deflow_amd.train train_data=<dir> sync_bn=true` (both ranks on this GPU,
    dist_backend=gloo): the ranks read disjoint shards of the index, step in lockstep and finish; the checkpoint loads"""
    import json
    import socket
    import subprocess
    import sys
    import deflow_amd
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    data = os.path.join(root, "tests", "golden", "av2_mini", "train")
    ck = str(tmp_path / "ddp.ckpt")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), "-m", "deflow_amd.train", "model=deflow", "lr=2e-4", "epochs=1", "batch_size=4",
                        "loss_fn=deflowLoss", "model.target.num_iters=2", "voxel_size=[0.4, 0.4, 6]", f"train_data={data}",
                        f"val_data={data}", "num_workers=2", "sync_bn=true", "gradient_clip_val=5.0", "dist_backend=gloo", "log_every=2",
                        f"save_checkpoint={ck}"], cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    steps = [l for l in lines if "trainer/loss" in l]
    assert len(steps) == ((95 + 1) // 2 // 4) // 2 and all(np.isfinite(l["trainer/loss"]) for l in steps)   # 48 items per rank, 12 steps
    assert any("val" in l for l in lines)
    m = deflow_amd.DeFlow(voxel_size=[0.4, 0.4, 6], grid_feature_size=[256, 256], num_iters=2)
    res = m.load_from_checkpoint(ck)
    assert not res.missing_keys and not res.unexpected_keys


def test_resume_continues_the_run_exactly(dev, tmp_path, capsys):
    """checkpoint / resume: two epochs in one run == one epoch, checkpoint, `resume=true` for the second epoch -- the same
    parameters bit for bit (weights, Adam moments and step count, epoch counter and the per-epoch data order all restored)"""
    import deflow_amd
    from deflow_amd import train as T
    base = ["model=deflow", "lr=2e-4", "batch_size=2", "loss_fn=deflowLoss", "model.target.num_iters=2",
            "voxel_size=[0.2, 0.2, 6]", "point_cloud_range=[-6.4, -6.4, -3, 6.4, 6.4, 3]", "pairs_per_epoch=6", "points_per_cloud=1200"]
    a, b1, b2 = (str(tmp_path / n) for n in ("a.ckpt", "b1.ckpt", "b2.ckpt"))
    T.main(base + ["epochs=2", f"save_checkpoint={a}"])
    T.main(base + ["epochs=1", f"save_checkpoint={b1}"])
    T.main(base + ["epochs=2", f"checkpoint={b1}", "resume=true", f"save_checkpoint={b2}"])
    capsys.readouterr()
    sa, sb = torch.load(a, weights_only=False), torch.load(b2, weights_only=False)
    assert sa["global_step"] == sb["global_step"] == 6 and sa["epoch"] == sb["epoch"] == 1
    for k, v in sa["state_dict"].items():
        assert torch.equal(v, sb["state_dict"][k]), k
    assert torch.equal(sa["optimizer_states"][0]["exp_avg"], sb["optimizer_states"][0]["exp_avg"])


def test_bench_line_contract(dev):
    """`python bench.py` (2 steps) prints ONE JSON line carrying what the round contract and the verdict ask for: metric /
    value / n_gpus / steps / warmup / ms_per_step consistent with each other, `roofline` with a frac in (0, 1], the
    HBM-bound stages, the forward-only and bf16 blocks -- so a broken bench is caught by the test run, not by the driver."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-loader"],
                       capture_output=True, text=True, cwd=root, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "frame-pairs/s" and d["dtype"] == "f32"
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert abs(d["value"] - 16 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and 0.3 < rf["frac"] <= 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert (rf["kernel"].startswith("conv_") or rf["kernel"].startswith("wgrad")) and rf["avg_launch_ms"] > 0
    # round 4: the dominant kernel is chosen among ALL MFMA kernels, the fp16x2 family is priced as a whole, the
    # sustained 16-bit MFMA rate of this chip is measured in the same run, and the precision ladder stands beside the headline
    assert rf["kernel"] == max(rf["all_mfma_kernels"], key=lambda k: rf["all_mfma_kernels"][k]["ms_per_step"])
    assert 0.2 < rf["fp16x2_family"]["frac"] < 1.0 and 0.3 < rf["fp16x2_family"]["share_of_step"] < 0.8
    su = rf["sustained"]
    assert 2000 < su["mfma_16bit_tflops_zero_operands"] < 2600 and 1000 < su["mfma_16bit_tflops_random_operands"] < su["mfma_16bit_tflops_zero_operands"]
    assert 1.2 < su["sustained_clock_ghz"] < 2.45 and rf["frac"] < su["frac_at_sustained_clock"] < 1.0
    for leg in ("strict_fp32", "gru_fp32"):
        assert "error" not in d[leg], d[leg]
        assert d[leg]["ms_per_step"] > d["ms_per_step"] * 0.98 and abs(d[leg]["loss"] - d["config"]["loss"]) <= 1e-4 * abs(d["config"]["loss"])
    assert d["strict_fp32"]["ms_per_step"] > d["gru_fp32"]["ms_per_step"]
    for k in ("pillarise_fwd", "bn_gelu_apply", "bn_gelu_bwd", "gru_fwd", "gru_bwd", "gru_wgrad", "pillarise_fwd_inference_b16"):
        assert k in d["roofline_hbm"] and 0 < d["roofline_hbm"][k]["frac"] < 1.2, k
    assert d["forward_only"]["ms_per_pair"] > 0 and d["bf16_inference"]["ms_per_pair"] > 0
    assert d["bf16_training"]["speedup_vs_fp32"] > 1.2 and d["bf16_training"]["configs4_shape"]["bf16_ms_per_step"] > 0
    assert d["hip_graph"]["host_ms_per_step"] < 20.0 and d["hip_graph"]["ms_per_step"] < 1.25 * d["hip_graph"]["eager_ms_per_step"]   # (wide: loaded hosts of the pool)
    assert "cpu_baseline" not in d   # --no-cpu-baseline


@pytest.mark.slow
def test_bench_hung_extra_leg_still_prints_the_headline(dev):
    """at N > 1 every extra leg behind the timed region runs under a wall-clock budget; a leg that hangs (test hook:
    the one-bucket leg sleeps forever on both ranks) ends the run with the HEADLINE line, `extras_aborted` naming the leg -- not with
    the driver's timeout and no line at all."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DF_BENCH_SHARE_GPU="1", DF_BENCH_TEST_HANG="one_bucket", DF_BENCH_LEG_BUDGET="20")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2"],
                       capture_output=True, text=True, cwd=root, timeout=600, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:] + r.stderr[-2500:]
    d = json.loads(lines[0])
    import bench as _bench
    assert d["metric"] == _bench.METRIC and d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["workload"] == _bench.WORKLOAD
    assert d["extras_aborted"]["leg"] == "one_bucket" and "allreduce_buckets" in d["extras_aborted"]["legs_completed"]


@pytest.mark.slow
def test_bench_two_ranks_share_the_gpu(dev):
    """`python bench.py --gpus 2` started bare -- the way the driver starts it -- with the real kernels: the file launches its own
    two ranks, both on this box's one GPU with gloo carrying the collectives (DF_BENCH_SHARE_GPU test hook; RCCL refuses two ranks
    on one device), and rank 0 prints the one JSON line with the N > 1 fields.  Losses of the ranks' shards stay finite, both ranks
    report, value = 2 x 16 pairs per step."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DF_BENCH_SHARE_GPU="1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, cwd=root, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["config"]["global_batch"] == 32 and d["config"]["parallelism"] == "dp2"
    assert abs(d["value"] - 32 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    assert d["rccl_ranks"] == [0, 0] and "gloo" in d["collective_backend"]
    # first-contact self-check: both ranks trained on their own shards, the arena is bit-identical after the warm-up
    assert d["rccl_selfcheck"]["params_bit_identical_across_ranks_after_warmup"] is True and d["rccl_selfcheck"]["steps_checked"] == 1
    assert 0 < d["per_rank_ms_per_step"]["min"] <= d["per_rank_ms_per_step"]["max"] <= d["ms_per_step"] * 1.05
    assert "allreduce_exposed_ms" in d and math.isfinite(d["allreduce_exposed_ms"])
    # first-8-GPU-run diagnostics: the collectives of one step bucket by bucket, and the one-bucket fallback leg
    bk = d["allreduce_buckets"]
    assert isinstance(bk, list) and len(bk) >= 4 and all(b["bytes"] > 0 and b["completed_ms"] >= b["issued_ms"] for b in bk), bk
    assert 0.9 * d["collectives"]["arena_bytes"] <= sum(b["bytes"] for b in bk) <= d["collectives"]["arena_bytes"]
    assert d["one_bucket"]["ms_per_step"] > 0 and d["one_bucket"]["bytes"] == d["collectives"]["arena_bytes"] and d["collectives"]["ranks"] == 2
    import bench as _bench
    assert d["metric"] == _bench.METRIC and d["config"]["workload"] == _bench.WORKLOAD
    assert math.isfinite(d["config"]["loss"]) and math.isfinite(d["bf16_training"]["loss"])
    assert d["bf16_training"]["pairs_per_s"] > 0      # (two ranks SHARING one GPU, the bf16 leg with the kernel profiler on: not a measurement)
    assert (d["roofline"]["kernel"].startswith("conv_") or d["roofline"]["kernel"].startswith("wgrad")) and "cpu_baseline" not in d and "forward_only" not in d
    # the captured data-parallel program: graph segments split at the buckets, host cost of a replay
    hg = d["hip_graph"]
    assert "error" not in hg, hg
    assert hg["graph_segments"] >= 6 and hg["allreduce_calls"] >= 6 and hg["host_ms_per_step"] < 20.0 and hg["ms_per_step"] > 0
