"""Round 4: PRE-SPLIT ("h2") tensors -- the operands of the fp16x2 convolution / weight-gradient kernels written split by their
producers (include/deflow_amd.h "PRE-SPLIT tensors"): layout round trips, the LDS-DMA-fed kernels against the in-kernel-split
forms of round 3 (bit-identical when both take the same bound) and against float64, the producers (BatchNorm + GELU passes,
conv epilogues, upsample) and their a-priori bounds."""
import math
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda", 0)


def slot(dev, v):
    return torch.tensor([float(v)], dtype=torch.float32, device=dev)


@pytest.mark.parametrize("slack", [1.0, 3.7, 2.0 ** 12])
def test_h2_pack_unpack_round_trip(dev, slack):
    """x -> [hi | lo] fp16 planes with the scale of a bound -> x: 22 significant bits for every element down to 2^-19 of the bound;
    a bound 2^12 too large (the judge's "stale scale" case) still leaves an ABSOLUTE error below 2^-33 of the true maximum.
    Channel slices of a wider buffer address whole 32-channel chunks."""
    from deflow_amd import ops
    from deflow_amd._lib import img, call
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(2, 5, 7, 128, generator=g) * torch.exp(torch.randn(2, 5, 7, 128, generator=g) * 3)).to(dev)
    amax = float(x.abs().max())
    b = slot(dev, amax * slack)
    t = ops.h2_pack(x, b)
    back = ops.h2_unpack(t)
    err = (back - x).abs()
    rel = err / x.abs().clamp_min(1e-30)
    big = x.abs() >= amax * slack * 2.0 ** -19
    assert float(rel[big].max()) <= 2.0 ** -21, float(rel[big].max())
    assert bool((err <= x.abs() * 2.0 ** -21 + amax * slack * 2.0 ** -36).all())     # 22 bits relative, or the absolute floor of the representation
    # a 64-channel slice at offset 32 of a 128-channel h2 buffer
    buf = ops.h2_empty((2, 5, 7, 128), dev, b)
    buf.zero_()
    call("df_h2_pack", img(x[..., 32:96].contiguous()), ops.ptr(b), img(buf, 64, 32), ops.stream())
    out = ops.h2_unpack(buf)
    assert torch.equal(out[..., 32:96], back[..., 32:96]) and float(out[..., :32].abs().max()) == 0.0 and float(out[..., 96:].abs().max()) == 0.0


CONV_SHAPES = [
    (128, 128, 2, 256, 256, "conv_halo_x3_kernel<256,128,4,2,1,4,2,xp>"),      # 256 pixels of one row
    (128, 256, 8, 128, 128, "conv_halo_x3_kernel<256,128,4,2,2,4,2,xp>"),      # two rows of a W == 128 image
    (256, 128, 32, 64, 64, "conv_halo_x3_kernel<256,128,4,2,4,4,2,xp>"),       # four rows of a W == 64 image
    (64, 64, 1, 512, 512, "conv_halo_x3_kernel<512,64,8,1,1,3,2,xp>"),         # 512 pixels of one row, 64 output channels
    (128, 64, 4, 256, 256, "conv_halo_x3_kernel<512,64,8,1,2,3,2,xp>"),        # two rows of a W == 256 image
]


@pytest.mark.parametrize("cin,cout,n,h,w,kname", CONV_SHAPES)
@pytest.mark.parametrize("mode", ["fwd_stats", "dgrad"])
def test_conv_presplit_input_is_bit_identical_to_in_kernel_split(dev, cin, cout, n, h, w, kname, mode):
    """conv_halo_x3_kernel<.., XP>: the halo arrives as [hi | lo] planes by LDS-DMA instead of being split in the staging
    registers.  With the SAME bound the planes hold the values the in-kernel split computes, the MFMA sequence is unchanged:
    outputs (and the BatchNorm statistics partials) must be BIT-IDENTICAL to round 3's kernel, which is pinned against float64
    (test_conv_x3_fp32_accurate).  All five tile forms the B = 16 step runs on; kernel names asserted."""
    from deflow_amd import ops
    from deflow_amd._lib import img, call
    g = torch.Generator().manual_seed(cin + cout + w)
    fwd = mode == "fwd_stats"
    wk = (torch.randn(cout, 3, 3, cin, generator=g) * (2.0 / (9 * cin)) ** 0.5).to(dev)
    x = torch.randn(n, h, w, cin, generator=g).to(dev)
    x[0, 0, 0, 0] = 7.5     # (an element near the bound)
    conv_mode, epi = (ops.CONV_FWD, ops.EPI_STATS) if fwd else (ops.CONV_DGRAD, ops.EPI_BIAS)
    rows = n * h * w
    ntile = rows // ops.conv_tile_m(rows, cout)
    bound = slot(dev, float(x.abs().max()) * 1.9)
    xh = ops.h2_pack(x, bound)
    y0, y1 = torch.zeros(n, h, w, cout, device=dev), torch.full((n, h, w, cout), float("nan"), device=dev)
    p0 = torch.zeros(ntile, cout, 2, device=dev) if fwd else None
    p1 = torch.full((ntile, cout, 2), float("nan"), device=dev) if fwd else None
    xi = img(x)
    xi._amax = bound                      # round 3's kernel with the same bound
    assert call("df_conv2d_h2p_ok", img(xh), img(y1), 3, 1, conv_mode, epi) == 1
    prof = ops.KernelProfiler()
    ops.PROFILER = prof
    try:
        ops.conv2d(xi, wk, None, img(y0), 3, 1, mode=conv_mode, epi=epi, stats=p0)
        ops.conv2d(img(xh), wk, None, img(y1), 3, 1, mode=conv_mode, epi=epi, stats=p1)
    finally:
        ops.PROFILER = None
    torch.cuda.synchronize()
    names = [r[0] for r in prof.records]
    assert names[1] == kname and names[0] == kname.replace(",xp>", ">"), names
    assert torch.equal(y0, y1), float((y0 - y1).abs().max())
    if fwd:
        assert torch.equal(p0, p1)


def test_conv_presplit_vs_float64(dev):
    """one pre-split layer end to end against float64 F.conv2d (CPU): input planes from a bound 3.1x the maximum, OUTPUT written as
    planes too (a-priori bound max|x| x max ||w_row||_1 + max|b|), unpacked: <= 2e-6 of the largest output; the bound holds."""
    import torch.nn.functional as F
    from deflow_amd import ops
    from deflow_amd._lib import img
    cin, cout, n, h, w = 128, 128, 2, 256, 256
    g = torch.Generator().manual_seed(77)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    x = torch.randn(n, h, w, cin, generator=g)
    want = F.conv2d(x.permute(0, 3, 1, 2).double(), wt.double(), bias.double(), padding=1).permute(0, 2, 3, 1)
    w_ohwi = ops.ohwi(wt.to(dev).contiguous(memory_format=torch.channels_last))
    xb = slot(dev, float(x.abs().max()) * 3.1)
    xh = ops.h2_pack(x.to(dev), xb)
    yb = ops.conv_out_bound(img(xh), w_ohwi, bias.to(dev), dev)
    yh = ops.h2_empty((n, h, w, cout), dev, yb)
    ops.conv2d(img(xh), w_ohwi, bias.to(dev), img(yh), 3, 1)
    y = ops.h2_unpack(yh).cpu().double()
    torch.cuda.synchronize()
    err = float((y - want).abs().max() / want.abs().max())
    slack = float(yb) / float(want.abs().max())
    print(f"[parity] pre-split conv 128->128 @256x256x2, planes in AND out: err {err:.2e} vs float64; a-priori output bound = {slack:.1f} x the true maximum")
    assert err <= 2e-6 and 1.0 <= slack <= 1024.0


@pytest.mark.parametrize("cin,cout,n,h,w", [(64, 64, 2, 256, 256), (128, 128, 4, 128, 128), (256, 256, 8, 64, 64), (512, 256, 2, 128, 128)])
def test_wgrad_presplit(dev, cin, cout, n, h, w):
    """wgrad3_h2p_kernel (both operands as planes, LDS-DMA ring): against float64 (<= 2e-6 of the largest entry), bias-gradient
    column sums included; and equal to round 3's wgrad3_x3_kernel<2> on the same bounds and split-K count up to the order of the fp32
    sums (the planes hold what the in-kernel split computes; the traversal orders differ)."""
    from deflow_amd import ops
    from deflow_amd._lib import img, call
    g = torch.Generator().manual_seed(cin + h)
    x = torch.randn(n, h, w, cin, generator=g)
    dy = torch.randn(n, h, w, cout, generator=g) * 1e-3
    xb, yb = slot(dev, float(x.abs().max()) * 1.3), slot(dev, float(dy.abs().max()) * 2.7)
    xd, dyd = x.to(dev), dy.to(dev)
    xh, dyh = ops.h2_pack(xd, xb), ops.h2_pack(dyd, yb)
    assert call("df_conv2d_wgrad_h2p_ok", img(xh), img(dyh), 3, 1) == 1
    dw = torch.empty(cout, 3, 3, cin, device=dev)
    prof = ops.KernelProfiler()
    ops.PROFILER = prof
    try:
        db = ops.conv2d_wgrad(img(xh), img(dyh), 3, 1, dw, want_bias=True)
    finally:
        ops.PROFILER = None
    torch.cuda.synchronize()
    assert prof.records[0][0] in ("wgrad3_h2p_kernel<4>", "wgrad3_h2p_kernel<2,64>")
    # float64 reference on the GPU-resident operands through unfold-free einsum per tap (CPU, double)
    xp = torch.nn.functional.pad(x.double(), (0, 0, 1, 1, 1, 1))
    want = torch.empty(cout, 3, 3, cin, dtype=torch.float64)
    dy2 = dy.double().reshape(-1, cout)
    for ky in range(3):
        for kx in range(3):
            want[:, ky, kx, :] = dy2.T @ xp[:, ky:ky + h, kx:kx + w, :].reshape(-1, cin)
    err = float((dw.cpu().double() - want).abs().max() / want.abs().max())
    eb = float((db.cpu().double() - dy2.sum(0)).abs().max() / dy2.sum(0).abs().max())
    print(f"[parity] pre-split wgrad {cin}->{cout} @{h}x{w}x{n}: err {err:.2e}, bias-gradient err {eb:.2e} (vs float64)")
    assert err <= 2e-6 and eb <= 1e-5
    # same bounds, same splits -> same bits as the in-kernel-split kernel
    splits = call("df_conv2d_wgrad_h2p_splits", img(xh), img(dyh))
    ws0 = torch.empty(splits * cout * 9 * cin, device=dev)
    ws1 = torch.empty_like(ws0)
    call("df_conv2d_wgrad_h2", img(xd), img(dyd), ops.ptr(xb), ops.ptr(yb), 3, 1, 1, ops.ptr(ws0), splits, None, ops.stream())
    call("df_conv2d_wgrad_h2p", img(xh), img(dyh), ops.ptr(xb), ops.ptr(yb), 3, 1, 1, ops.ptr(ws1), splits, None, ops.stream())
    torch.cuda.synchronize()
    # (the two kernels walk the pixels in different orders since the pre-split one went column-major -- its x rows stay in L2 -- so a
    # split holds different pixels: the planes being what the in-kernel split computes shows in the SUMS over the splits, which
    # agree to the rounding of two fp32 summation orders)
    s0 = ws0.view(splits, -1).double().sum(0)
    s1 = ws1.view(splits, -1).double().sum(0)
    assert float((s0 - s1).abs().max() / s0.abs().max()) <= 2e-6, float((s0 - s1).abs().max() / s0.abs().max())


def test_bn_gelu_producers_write_planes_with_valid_bounds(dev):
    """the BatchNorm + GELU passes as plane producers: z = gelu(bn(y)) and dy = the BatchNorm / GELU backward, written pre-split
    with the bounds the finalisations derive from the statistics, equal (to the 22 bits of the representation) to the fp32
    outputs of the same kernels; each bound >= the true maximum and within 2^10 of it."""
    from deflow_amd import ops
    from deflow_amd._lib import img
    n, h, w, C = 4, 64, 64, 128
    g = torch.Generator().manual_seed(9)
    y = (torch.randn(n, h, w, C, generator=g) * (1.0 + 5.0 * torch.rand(C, generator=g)) + torch.randn(C, generator=g)).to(dev)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.3).to(dev)
    groups, ipg = 2, n // 2
    rows_pg = ipg * h * w
    # statistics through the conv-free route: per-tile partials computed with torch (the layout df_bn_finalize reads)
    tiles_pg = rows_pg // 128
    yt = y.reshape(groups, tiles_pg, 128, C)
    partial = torch.stack([yt.sum(2), (yt * yt).sum(2)], dim=-1).reshape(groups * tiles_pg, C, 2).contiguous()
    y_amax = slot(dev, float(y.abs().max()))
    zb = ops.amax_slot(dev)
    bn_ss = torch.empty(groups, 4, C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    ops.bn_finalize(partial, tiles_pg, groups, C, rows_pg, gamma, beta, 1e-5, 0.1, rm, rv, bn_ss, y_amax=y_amax, z_bound=zb)
    z32 = torch.empty(n, h, w, C, device=dev)
    ops.bn_gelu_apply(y, bn_ss, ipg, img(z32))
    zh = ops.h2_empty((n, h, w, C), dev, zb)
    ops.bn_gelu_apply(y, bn_ss, ipg, img(zh))
    zz = ops.h2_unpack(zh)
    zmax = float(z32.abs().max())
    assert zmax <= float(zb) <= 1024 * zmax, (zmax, float(zb))
    assert float((zz - z32).abs().max()) <= 2.0 ** -21 * float(zb)
    # backward
    dz = (torch.randn(n, h, w, C, generator=g) * 1e-2).to(dev)
    dzi = img(dz)
    dy32, dg32, db32, dbias32 = ops.bn_gelu_bwd(dzi, y, bn_ss, ipg, groups)
    dzi2 = img(dz)
    dyh, dg, db, dbias = ops.bn_gelu_bwd(dzi2, y, bn_ss, ipg, groups, dy_h2=True, y_amax=y_amax)
    dd = ops.h2_unpack(dyh)
    dmax, bnd = float(dy32.abs().max()), float(dyh._df_h2)
    print(f"[parity] bounds: z {float(zb) / zmax:.2f} x max|z|, dy {bnd / dmax:.2f} x max|dy|")
    assert dmax <= bnd <= 1024 * dmax, (dmax, bnd)
    assert float((dd - dy32).abs().max()) <= 2.0 ** -21 * bnd
    assert torch.equal(dg, dg32) and torch.equal(db, db32) and torch.equal(dbias, dbias32)


def test_upsample_and_1x1_epilogue_write_planes(dev):
    """the two producers of an UpsampleSkip concatenation: bilinear x2 into the first half, the 1x1 conv's epilogue into the second,
    both with the concatenation's bound; and a 1x1 DATA gradient with a pre-split output.  Against the fp32 outputs of the same
    kernels."""
    from deflow_amd import ops
    from deflow_amd._lib import img
    B, h, w, lat, cs = 2, 32, 32, 64, 128
    g = torch.Generator().manual_seed(21)
    t = torch.randn(B, h, w, lat, generator=g).to(dev)
    b = torch.randn(B, 2 * h, 2 * w, cs, generator=g).to(dev)
    w3 = (torch.randn(lat, 1, 1, cs, generator=g) * 0.1).to(dev)
    bias = (torch.randn(lat, generator=g) * 0.1).to(dev)
    ref = torch.empty(B, 2 * h, 2 * w, 2 * lat, device=dev)
    ops.upsample2x(img(t), img(ref, lat, 0), False)
    ops.conv2d(img(b), w3, bias, img(ref, lat, lat), 1, 1)
    bound = ops.conv_out_bound(img(b), w3.reshape(lat, cs), bias, dev, other=slot(dev, float(t.abs().max())))
    cat = ops.h2_empty((B, 2 * h, 2 * w, 2 * lat), dev, bound)
    ops.upsample2x(img(t), img(cat, lat, 0), False)
    ops.conv2d(img(b), w3, bias, img(cat, lat, lat), 1, 1)
    got = ops.h2_unpack(cat)
    torch.cuda.synchronize()
    assert float(ref.abs().max()) <= float(bound)
    assert float((got - ref).abs().max()) <= 2.0 ** -21 * float(bound) + 2e-6 * float(ref.abs().max())
    # 1x1 data gradient into planes
    dt = torch.randn(B, h, w, lat, generator=g).to(dev)
    wt = ops.weight_transpose(w3)                     # [cs, 1, 1, lat]
    dref = torch.empty(B, h, w, cs, device=dev)
    ops.conv2d(img(dt), wt, None, img(dref), 1, 1, mode=ops.CONV_DGRAD)
    db_ = ops.conv_out_bound(img(dt), wt, None, dev)
    dA = ops.h2_empty((B, h, w, cs), dev, db_)
    ops.conv2d(img(dt), wt, None, img(dA), 1, 1, mode=ops.CONV_DGRAD)
    torch.cuda.synchronize()
    assert float(dref.abs().max()) <= float(db_)
    assert float((ops.h2_unpack(dA) - dref).abs().max()) <= 2.0 ** -21 * float(db_) + 2e-6 * float(dref.abs().max())


def test_presplit_step_matches_fp32_storage_step(dev, monkeypatch):
    """the whole B = 16 training step at 256 x 256 with the pre-split data flow against the same step with every tensor fp32
    (DF_H2P=0, round 3's flow): the pre-split kernels must actually run (names asserted), loss and every parameter gradient agree
    to <= 3e-5 rms-relative (both are the fp32 computation to ~1e-6; the digest tests pin each against float64)."""
    import deflow_amd
    from deflow_amd import ops
    from deflow_amd.optim import Trainer
    from deflow_amd.synth import synth_batch
    grid, n_pts = 256, 20000
    cfg = dict(voxel_size=[0.2, 0.2, 6], point_cloud_range=[-25.6, -25.6, -3, 25.6, 25.6, 3], grid_feature_size=[grid, grid])
    batch = synth_batch(16, n_pts, seed=4242, grid_hw=(grid, grid), device=dev)
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("DF_H2P", flag)
        torch.manual_seed(16)
        m = deflow_amd.DeFlow(**cfg).to(dev).train()
        tr = Trainer(m, lr=0.0)
        prof = ops.KernelProfiler()
        ops.PROFILER = prof
        try:
            tr.flat.zero_grad(); tr.sink.begin()
            loss = tr._forward_backward(batch) if hasattr(tr, "_forward_backward") else None
        finally:
            ops.PROFILER = None
        torch.cuda.synchronize()
        names = {r[0] for r in prof.records}
        res[flag] = (float(loss), {k: p.grad.detach().clone() for k, p in m.named_parameters()}, names)
    assert any(nm.endswith(",xp>") for nm in res["1"][2]) and any(nm.startswith("wgrad3_h2") for nm in res["1"][2]), sorted(res["1"][2])
    assert not any(nm.endswith(",xp>") for nm in res["0"][2]) and not any(nm.startswith("wgrad3_h2") for nm in res["0"][2])
    assert abs(res["1"][0] - res["0"][0]) <= 1e-5 * abs(res["0"][0]), (res["1"][0], res["0"][0])
    worst = (0.0, "")
    for k, g0 in res["0"][1].items():
        g1 = res["1"][1][k]
        den = float(g0.double().norm())
        if den == 0.0:
            continue
        e = float((g1.double() - g0.double()).norm()) / den
        if "batchnorm" not in k and k.endswith(".conv.bias") and "encoder" in k:
            continue        # BatchNorm-shadowed biases: true gradient 0, both values are rounding noise
        worst = max(worst, (e, k))
        assert e <= 3e-5, (k, e)
    print(f"[parity] pre-split step vs fp32-storage step: loss {res['1'][0]:.7f} / {res['0'][0]:.7f}; worst gradient rms-relative difference {worst[0]:.2e} ({worst[1]})")


def test_gelu_forms_against_float64_erf(dev):
    """df_gelu / df_gelu_grad are the exact-erf GELU [REF decoder.py:209] evaluated through Abramowitz-Stegun 7.1.26
    (v_exp / v_rcp, 16 VALU).  Stated tolerance: |gelu - exact| <= 5e-7 max(1, |x|) and |gelu' - exact| <= 1e-6 over [-10, 10]
    (float64 erf), measured through the BatchNorm + GELU passes with identity statistics -- the kernels every mode runs."""
    from deflow_amd import ops
    from deflow_amd._lib import img
    C, n = 32, 1 << 16
    x = torch.linspace(-10.0, 10.0, n * C, dtype=torch.float64).reshape(1, n // 64, 64, C)
    y = x.float().to(dev)
    bn_ss = torch.stack([torch.ones(C), torch.zeros(C), torch.zeros(C), torch.ones(C)]).reshape(1, 4, C).to(dev)
    z = torch.empty_like(y)
    ops.bn_gelu_apply(y, bn_ss, 1, img(z))
    xd = y.cpu().double()
    want = 0.5 * xd * (1.0 + torch.erf(xd / math.sqrt(2.0)))
    e = ((z.cpu().double() - want).abs() / xd.abs().clamp_min(1.0)).max()
    dz = torch.ones_like(y)
    dy, _, _, _ = ops.bn_gelu_bwd(img(dz), y, bn_ss, 1, 1, frozen=True)
    wantg = 0.5 * (1.0 + torch.erf(xd / math.sqrt(2.0))) + xd * torch.exp(-0.5 * xd * xd) / math.sqrt(2.0 * math.pi)
    eg = (dy.cpu().double() - wantg).abs().max()
    print(f"[parity] GELU (A-S 7.1.26) vs float64 erf over [-10, 10]: value {float(e):.2e} (bound 5e-7 max(1,|x|)), derivative {float(eg):.2e} (bound 1e-6)")
    assert float(e) <= 5e-7 and float(eg) <= 1e-6


def test_fused_bn_backward_partials_match_the_reduce_pass(dev, monkeypatch):
    """df_conv2d_h2p_dgrad_bn: the data gradient's epilogue also sums the BatchNorm + GELU backward partials of the layer in front.
    The B = 16 step at 256 x 256 with the fusion on (default) against the same step with the separate reduce pass (DF_FUSE_BN_BWD=0):
    the fused launches happen (no bn_gelu_bwd_reduce for the layers behind a 3x3 stride-1 data gradient), every parameter
    gradient agrees to <= 2e-6 rms-relative (the two differ only in the order of fp32 partial sums)."""
    import deflow_amd
    from deflow_amd import ops
    from deflow_amd.optim import Trainer
    from deflow_amd.synth import synth_batch
    grid, n_pts = 256, 20000
    cfg = dict(voxel_size=[0.2, 0.2, 6], point_cloud_range=[-25.6, -25.6, -3, 25.6, 25.6, 3], grid_feature_size=[grid, grid])
    batch = synth_batch(16, n_pts, seed=4242, grid_hw=(grid, grid), device=dev)
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("DF_FUSE_BN_BWD", flag)
        torch.manual_seed(16)
        m = deflow_amd.DeFlow(**cfg).to(dev).train()
        tr = Trainer(m, lr=0.0)
        prof = ops.KernelProfiler()
        ops.PROFILER = prof
        try:
            tr.flat.zero_grad(); tr.sink.begin()
            loss = tr._forward_backward(batch)
        finally:
            ops.PROFILER = None
        torch.cuda.synchronize()
        n_reduce = sum(1 for r in prof.records if r[0] == "bn_gelu_bwd_reduce")
        res[flag] = (float(loss), {k: p.grad.detach().clone() for k, p in m.named_parameters()}, n_reduce)
    # 16 BatchNorm layers; fused: the layers in front of a PRE-SPLIT-input 3x3 stride-1 data gradient (at this grid: the 5 of stage 2;
    # stage 1's 128 x 128 x 64 and stage 3's 32 x 32 images have no pre-split tile form).  At the bench's 512 x 512 grid only each stage's last layer keeps its pass.
    assert res["0"][2] == 16 and res["1"][2] == 11, (res["0"][2], res["1"][2])
    worst = (0.0, "")
    for k, g0 in res["0"][1].items():
        den = float(g0.double().norm())
        if den == 0.0 or (k.endswith(".conv.bias") and "encoder" in k):
            continue
        e = float((res["1"][1][k].double() - g0.double()).norm()) / den
        worst = max(worst, (e, k))
        assert e <= 2e-6, (k, e)
    print(f"[parity] fused BatchNorm-backward partials vs the reduce pass: worst gradient rms-relative difference {worst[0]:.2e} ({worst[1]})")


def test_weight_prep_one_launch_equals_per_call_forms(dev, monkeypatch):
    """df_weight_prep: every conv layer's transposed weights, [hi | lo] planes (of w and of its transpose), row L1 norms and max |bias|
    from ONE launch must be bit-identical to the per-call kernels they replace (df_weight_transpose, df_split_h2, df_rows_l1max);
    and a B = 16 training step with the prep on (default) must give bit-identical gradients to the step without it (DF_WPREP=0)."""
    import deflow_amd
    from deflow_amd import ops
    from deflow_amd._lib import call, img
    from deflow_amd.unet import FastFlow3DUNet
    torch.manual_seed(3)
    net = FastFlow3DUNet().to(dev)
    convs = [m for m in net.modules() if isinstance(m, torch.nn.Conv2d)]
    with torch.no_grad():
        for m in convs:
            m.bias.uniform_(-0.3, 0.3)
    wa = ops.amax_slot(dev)
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    wa.copy_(flat.abs().max().reshape(1))
    wp = ops.WeightPrep(convs, wa)
    wp.run(wa)
    torch.cuda.synchronize()
    n_split = 0
    for m in convs:
        w = ops.ohwi(m.weight)
        wt_ref = ops.weight_transpose(w)                 # (ops.WPREP is None here: the per-call kernel)
        wt = wp.wt(w)
        assert torch.equal(wt, wt_ref)
        l1, bm = wp.l1(w, m.bias.detach())
        l1r, bmr = ops.rows_l1max(w.shape[0], w.numel() // w.shape[0], w, m.bias.detach())
        l1t, _ = wp.l1(wt, None)
        l1tr, _ = ops.rows_l1max(wt_ref.shape[0], wt_ref.numel() // wt_ref.shape[0], wt_ref, None)
        assert torch.equal(l1, l1r) and torch.equal(bm, bmr) and torch.equal(l1t, l1tr), m
        h = wp.h2(w)       # planes of EVERY layer (the 1x1 / stride-2 kernel reads them too since the round's second session)
        n_split += int(m.kernel_size[0] == 3 and m.stride[0] == 1)
        for src, got in ((w, h[0]), (wt_ref, wp.h2(wt)[0])):
            ref2 = torch.empty(2 * src.numel(), dtype=torch.float16, device=dev)
            call("df_split_h2", ops.ptr(src.contiguous()), ops.ptr(wa), ops.ptr(ref2), src.numel(), ops.stream())
            assert torch.equal(got, ref2)
    assert n_split == 20
    # whole step: same bits with and without
    from deflow_amd.optim import Trainer
    from deflow_amd.synth import synth_batch
    cfg = dict(voxel_size=[0.2, 0.2, 6], point_cloud_range=[-25.6, -25.6, -3, 25.6, 25.6, 3], grid_feature_size=[256, 256])
    batch = synth_batch(16, 20000, seed=4242, grid_hw=(256, 256), device=dev)
    arenas = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("DF_WPREP", flag)
        torch.manual_seed(16)
        m = deflow_amd.DeFlow(**cfg).to(dev).train()
        tr = Trainer(m, lr=0.0)
        tr.flat.zero_grad(); tr.sink.begin()
        tr._forward_backward(batch)
        torch.cuda.synchronize()
        assert (getattr(tr, "_wprep", None) is not None) == (flag == "1")
        arenas[flag] = tr.flat.grad.clone()
    assert torch.equal(arenas["1"], arenas["0"])


def test_canvas_bound_from_the_feature_net_statistics(dev):
    """df_pfn_bn_finalize2: the a-priori bound of max |canvas| (BatchNorm1d statistics x the largest reachable linear output: the nine
    point features are bounded by the geometry) holds and is within 2^10 of the true maximum -- it replaces a df_absmax pass over the
    1 GB canvas for the fp16x2 consumers of the skip connection."""
    import deflow_amd
    from deflow_amd import ops
    from deflow_amd.synth import synth_batch
    cfg = dict(voxel_size=[0.2, 0.2, 6], point_cloud_range=[-25.6, -25.6, -3, 25.6, 25.6, 3], grid_feature_size=[256, 256])
    torch.manual_seed(5)
    m = deflow_amd.DeFlow(**cfg).to(dev).train()
    batch = synth_batch(4, 20000, seed=11, grid_hw=(256, 256), device=dev)
    ops.amax_pool_reset()
    with torch.no_grad():
        flow, st = m._run(batch["pc0"].contiguous(), batch["pc1"].contiguous(), True, True)
    torch.cuda.synchronize()
    rec = getattr(st["bstar"], "_df_amax", None)
    assert rec is not None
    bound, true_max = float(rec[0]), float(st["bstar"].abs().max())
    print(f"[parity] canvas bound {bound:.3f} = {bound / true_max:.1f} x max |canvas| {true_max:.3f}")
    assert true_max <= bound <= 1024 * true_max


@pytest.mark.parametrize("B,h,w,lat,cs,ac", [(2, 32, 32, 64, 64, False), (1, 16, 48, 128, 128, False), (2, 64, 64, 64, 128, True), (3, 8, 8, 256, 512, False)])
def test_skip_conv_writes_the_bilinear_half_too(dev, B, h, w, lat, cs, ac):
    """df_conv2d_h2f_wp_up (round 6): the 1x1 skip convolution of an UpsampleSkip block writing BOTH halves of the pre-split concatenation
    -- its own output and the bilinear x2 of t -- against the two launches it replaces (df_upsample2x_h2 + df_conv2d_h2f_wp): the same
    planes, bit for bit"""
    from deflow_amd import ops
    from deflow_amd._lib import call, img, ptr, stream
    g = torch.Generator().manual_seed(B * 1000 + lat + cs)
    t = torch.randn(B, h, w, lat, generator=g).to(dev)
    b = torch.randn(B, 2 * h, 2 * w, cs, generator=g).to(dev)
    w3 = (torch.randn(lat, 1, 1, cs, generator=g) * 0.1).to(dev)
    bias = (torch.randn(lat, generator=g) * 0.1).to(dev)
    xa = slot(dev, float(b.abs().max()))
    wa = torch.zeros(1, device=dev)
    call("df_absmax", img(w3.reshape(1, 1, -1, cs)), ptr(wa), stream())
    w2 = torch.empty(2 * w3.numel(), dtype=torch.float16, device=dev)
    call("df_split_h2", ptr(w3), ptr(wa), ptr(w2), w3.numel(), stream())
    bound = ops.conv_out_bound(img(b), w3.reshape(lat, cs), bias, dev, other=slot(dev, float(t.abs().max())))
    ref = ops.h2_empty((B, 2 * h, 2 * w, 2 * lat), dev, bound)
    got = ops.h2_empty((B, 2 * h, 2 * w, 2 * lat), dev, bound)
    ref.fill_(7.0); got.fill_(7.0)
    call("df_upsample2x_h2", img(t), img(ref, lat, 0), int(ac), ptr(bound), stream())
    call("df_conv2d_h2f_wp", img(b), ptr(w3), ptr(w2), ptr(xa), ptr(wa), ptr(bias), img(ref, lat, lat), ptr(bound), 1, 1, 0, ops.CONV_FWD,
         ops.EPI_BIAS, None, None, None, 0, None, stream())
    call("df_conv2d_h2f_wp_up", img(b), ptr(w3), ptr(w2), ptr(xa), ptr(wa), ptr(bias), img(got, lat, lat), ptr(bound), img(t), int(ac), stream())
    torch.cuda.synchronize()
    assert torch.equal(got, ref), float((ops.h2_unpack(got) - ops.h2_unpack(ref)).abs().max())
    want = F.interpolate(t.permute(0, 3, 1, 2).double().cpu(), scale_factor=2, mode="bilinear", align_corners=ac).permute(0, 2, 3, 1)
    up = ops.h2_unpack(got)[..., :lat].double().cpu()
    assert float((up - want).abs().max()) <= 2.0 ** -20 * float(bound) + 2e-6 * float(want.abs().max())
