"""FastFlow3DUNet backbone ([REF deflow.py:15,32,87-88]) built from ConvWithNorms ([REF decoder.py:202-220]).

Module tree / state_dict keys follow upstream (encoder_step_{1,2,3}.N.{conv,batchnorm}, decoder_step{1,2,3}.
{u1_u2.0,u3,u4_u5.{0,1}}, decoder_step4).  Compute is an explicit engine over HIP kernels (csrc/conv.hip, csrc/conv_wgrad.hip,
csrc/elementwise.hip) with hand-sequenced backward: NHWC activations, the two clouds batched through the
shared encoder as 2B images with two BatchNorm statistic groups (= the reference's two encoder calls), channel
concatenations realised by writing into slices of one buffer (no torch.cat copies), gradients that meet at a
concatenation summed by the producing kernel's `accumulate` epilogue.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn

from . import ops
from ._lib import DfImg, call, img, img_pair, ptr, stream, ver

import os
_NO_FUSED_BIAS = bool(int(os.environ.get("DF_NO_FUSED_BIAS", "0")))  # A/B switch: separate column-sum pass for conv bias grads
_SPARSE_H2 = os.environ.get("DF_SPARSE_H2", "1") != "0"   # A/B switch: the sparse edge kernels (last conv forward / weight gradient) as fp16x2 products


class ConvWithNorms(nn.Module):
    def __init__(self, in_num_channels: int, out_num_channels: int, kernel_size: int, stride: int, padding: int):
        super().__init__()
        self.conv = nn.Conv2d(in_num_channels, out_num_channels, kernel_size, stride, padding)
        self.batchnorm = nn.BatchNorm2d(out_num_channels)
        self.nonlinearity = nn.GELU()
        self.stride = stride

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """Stand-alone call on an NCHW tensor (inference or training statistics; no autograd)."""
        xh = x.permute(0, 2, 3, 1).contiguous()
        n, h, w, _ = xh.shape
        ho, wo = (h - 1) // self.stride + 1, (w - 1) // self.stride + 1
        z = torch.empty(n, ho, wo, self.conv.out_channels, dtype=torch.float32, device=x.device)
        with torch.no_grad():
            if ho == 1 and wo == 1:
                # [REF decoder.py:214-217]: BatchNorm is skipped on 1x1 maps (in training AND eval; no statistics update) ->
                # gelu(conv(x) + bias): the folded epilogue with scale 1, shift 0
                C = self.conv.out_channels
                one = torch.ones(C, dtype=torch.float32, device=x.device)
                ops.conv2d(img(xh), ops.ohwi(self.conv.weight), self.conv.bias.detach(), img(z), self.conv.kernel_size[0],
                           self.stride, epi=ops.EPI_BN_GELU, scale=one, shift=torch.zeros_like(one))
            else:
                _cwn_forward(self, img(xh), img(z), n, 1, self.training, None)
        return z.permute(0, 3, 1, 2)


class UpsampleSkip(nn.Module):
    def __init__(self, skip_channels: int, latent_channels: int, out_channels: int):
        super().__init__()
        self.u1_u2 = nn.Sequential(nn.Conv2d(skip_channels, latent_channels, 1, 1, 0), nn.Identity())
        self.u3 = nn.Conv2d(latent_channels, latent_channels, 1, 1, 0)
        self.u4_u5 = nn.Sequential(nn.Conv2d(2 * latent_channels, out_channels, 3, 1, 1),
                                   nn.Conv2d(out_channels, out_channels, 3, 1, 1))


def _cwn_forward(m: ConvWithNorms, x: DfImg, z: DfImg, n_imgs: int, groups: int, train: bool, tape: Optional[list],
                 store16: bool = False, h2_stage: bool = False):
    """z = gelu(bn(conv(x))).  train: batch statistics per group (+ running update); else running stats, fused.
    store16 (training, bf16-storage mode): the conv output y is kept as bfloat16 (z's type is the caller's: its descriptor)."""
    dev = m.conv.weight.device
    w, b, bn = ops.ohwi(m.conv.weight), m.conv.bias.detach(), m.batchnorm
    C = m.conv.out_channels
    if not train and tape is None:
        scale, shift, _, _ = ops.folded_bn(bn)
        ops.conv2d(x, w, b, z, 3, m.stride, epi=ops.EPI_BN_GELU, scale=scale, shift=shift)
        return
    if not train:
        # eval mode WITH a tape (fine-tuning with frozen BatchNorm, saliency, gradient checks: the reference module is
        # differentiable in eval mode): keep the conv output y and normalise with the running statistics through the same
        # two kernels as training, so the shared backward applies with the batch-statistic terms switched off
        y = torch.empty(n_imgs, z.h, z.w, C, dtype=torch.float32, device=dev)
        ops.conv2d(x, w, b, img(y), 3, m.stride, epi=ops.EPI_BIAS)
        bn_ss = torch.stack(list(ops.folded_bn(bn))).view(1, 4, C).contiguous()   # scale, shift, running mean, invstd
        ops.bn_gelu_apply(y, bn_ss, n_imgs, z)
        tape.append(("cwn", m, x, y, bn_ss, n_imgs, 1, True))
        return
    ipg = n_imgs // groups
    rows_pg = ipg * z.h * z.w
    tile_m = ops.conv_tile_m(rows_pg, C)
    assert rows_pg % tile_m == 0, "BatchNorm statistic groups must be a multiple of the row tile"
    tiles_pg = rows_pg // tile_m
    y = torch.empty(n_imgs, z.h, z.w, C, dtype=torch.bfloat16 if store16 else torch.float32, device=dev)
    partial = torch.empty(tiles_pg * groups, C, 2, dtype=torch.float32, device=dev)
    yi = img(y)
    yi.grp_size = ipg  # stat groups are image groups: tile -> group by its first row
    yi.grp_off = ipg * y.stride(0)
    # pre-split mode (h2_stage: this layer's z and / or the dy its backward produces are h2 images): the epilogue also measures
    # max |y|, from which the finalisations derive the bounds that scale z (here) and dy (ops.bn_gelu_bwd)
    ya = ops.amax_slot(dev) if (h2_stage and not store16) else None
    ops.conv2d(x, w, b, yi, 3, m.stride, epi=ops.EPI_STATS, stats=partial, amax_out=ya)
    bn_ss = torch.empty(groups, 4, C, dtype=torch.float32, device=dev)
    zb = z._amax if z.elt == 2 else None       # (zero-initialised slot: the h2 tensor was created with it)
    ops.bn_finalize(partial, tiles_pg, groups, C, rows_pg, bn.weight.detach(), bn.bias.detach(), bn.eps, bn.momentum,
                    bn.running_mean, bn.running_var, bn_ss, y_amax=ya if zb is not None else None, z_bound=zb)
    ops.bump_tracked(bn.num_batches_tracked, groups)
    ops.bn_gelu_apply(y, bn_ss, ipg, z)
    if ya is not None:
        y._df_yamax = ya
    if tape is not None:
        tape.append(("cwn", m, x, y, bn_ss, ipg, groups, False))


def _stage_store16_ok(n: int, h: int, w: int, c: int, dev) -> bool:
    """do the bf16-tile kernels (df_conv2d_w16 with bfloat16 tensors, df_conv2d_wgrad_bf16) exist for an [n,h,w,c] -> [n,h,w,c]
    3x3 stride-1 layer?  (asks the library with descriptors of that shape; nothing is launched)"""
    if os.environ.get("DF_BF16_STORE") == "0" or w % 32 or c % 64:
        return False
    probe = torch.empty(16, dtype=torch.bfloat16, device=dev)
    d = DfImg(probe.data_ptr(), n, h, w, c, c, n, h * w * c, 0, 1, 0)
    return (call("df_conv2d_w16_ok", d, d, 3, 1, ops.CONV_FWD, ops.EPI_STATS) == 1
            and call("df_conv2d_w16_ok", d, d, 3, 1, ops.CONV_DGRAD, ops.EPI_BIAS) == 1)


def _h2p_ok(n: int, h: int, w: int, cin: int, cout: int, dev) -> bool:
    """do the pre-split forms (df_conv2d_h2p with an h2 input: forward and data gradient; df_conv2d_wgrad_h2p) exist for an
    [n,h,w,cin] -> [n,h,w,cout] 3x3 stride-1 layer?  (descriptors of that shape; nothing is launched)"""
    if cin % 64 or cout % 64 or not ops.h2p_on():
        return False
    probe = torch.empty(64, dtype=torch.float32, device=dev)
    base = (probe.data_ptr() + 127) // 128 * 128

    def d(c, elt):
        return DfImg(base, n, h, w, c, c, n, h * w * c, 0, elt, 0)
    return (call("df_conv2d_h2p_ok", d(cin, 2), d(cout, 0), 3, 1, ops.CONV_FWD, ops.EPI_BIAS) == 1
            and call("df_conv2d_h2p_ok", d(cout, 2), d(cin, 0), 3, 1, ops.CONV_DGRAD, ops.EPI_BIAS) == 1
            and call("df_conv2d_wgrad_h2p_ok", d(cin, 2), d(cout, 2), 3, 1) == 1)


def _is_h2(t) -> bool:
    return getattr(t, "_df_h2", None) is not None


class FastFlow3DUNet(nn.Module):
    def __init__(self, align_corners: bool = False):
        super().__init__()
        C = ConvWithNorms
        self.encoder_step_1 = nn.Sequential(C(32, 64, 3, 2, 1), *[C(64, 64, 3, 1, 1) for _ in range(3)])
        self.encoder_step_2 = nn.Sequential(C(64, 128, 3, 2, 1), *[C(128, 128, 3, 1, 1) for _ in range(5)])
        self.encoder_step_3 = nn.Sequential(C(128, 256, 3, 2, 1), *[C(256, 256, 3, 1, 1) for _ in range(5)])
        self.decoder_step1 = UpsampleSkip(512, 256, 256)
        self.decoder_step2 = UpsampleSkip(256, 128, 128)
        self.decoder_step3 = UpsampleSkip(128, 64, 64)
        self.decoder_step4 = nn.Conv2d(64, 64, 3, 1, 1)
        self.align_corners = align_corners
        for p in self.parameters():  # OHWI memory so the kernels read weights with no per-step transform
            if p.dim() == 4:
                p.data = p.data.contiguous(memory_format=torch.channels_last)

    # ------------------------------------------------------------------------------- forward ----
    def run(self, bstar: torch.Tensor, train: bool, tape: Optional[list], out_cells=None) -> torch.Tensor:
        """bstar [B,H,W,64] = cat(pc0 canvas, pc1 canvas) -> [B,H,W,64].  Appends what backward needs to `tape`.
        out_cells: PillarState of the cloud whose occupied cells are the only pixels of the output anyone reads (the
        decoder gathers at pc0's cells): the last conv is then evaluated there only (df_sparse_conv3x3) and the rest of
        the returned tensor is NOT written."""
        B, H, W, _ = bstar.shape
        assert H % 8 == 0 and W % 8 == 0
        dev = bstar.device
        f32 = dict(dtype=torch.float32, device=dev)
        x = img_pair(bstar, 32)
        cats: List[torch.Tensor] = []
        alive: List[torch.Tensor] = []  # DfImg descriptors hold raw pointers: every layer output must outlive its readers
        h, w = H, W
        for stage in (self.encoder_step_1, self.encoder_step_2, self.encoder_step_3):
            for i, m in enumerate(stage):
                if i == 0:
                    h, w = h // 2, w // 2
                    # bf16-storage mode (ops.BF16_STORE, training with a tape): inside a stage whose 3x3 stride-1 layers have
                    # the bf16-tile kernels (W % 128 == 0, or W == 64) every y / z / dz / dy is bfloat16; the stage's input
                    # and its last activation (the skip tensor) stay fp32
                    store16 = bool(train and tape is not None and ops.BF16_STORE and ops.MFMA_BF16 and _stage_store16_ok(2 * B, h, w, m.conv.out_channels, dev))
                    # pre-split mode (fp32 training on the fp16x2 kernels, round 4): every activation BETWEEN the layers of a stage
                    # (z of layers 0 .. L-2: read only by the next 3x3 stride-1 conv and its weight gradient) and the dy of layers
                    # 1 .. L-1 are h2 images; the stage's input, its last activation (the skip tensor), y and dz stay fp32
                    h2s = bool(train and tape is not None and not store16 and _h2p_ok(2 * B, h, w, m.conv.out_channels, m.conv.out_channels, dev))
                C = m.conv.out_channels
                if i == len(stage) - 1:
                    cat = torch.empty(B, h, w, 2 * C, **f32)
                    cats.append(cat)
                    z = img_pair(cat, C)
                    keep = cat
                elif h2s:
                    keep = ops.h2_empty((2 * B, h, w, C), dev, ops.amax_slot(dev))
                    z = img(keep)
                else:
                    keep = torch.empty(2 * B, h, w, C, dtype=torch.bfloat16 if store16 else torch.float32, device=dev)
                    z = img(keep)
                _cwn_forward(m, x, z, 2 * B, 2, train, tape, store16, h2s)
                if i == len(stage) - 1 and getattr(z, "_amax", None) is not None:
                    cat._df_amax = (z._amax, ver(cat))      # the two clouds' images ARE the skip tensor: its readers inherit the bound
                alive.append(keep)  # without this a no-tape run would free x's tensor before the next conv reads it
                if tape is not None:
                    tape.append(("keep", keep))
                x = z
        fstar, lstar, rstar = cats
        s = self._upsample_skip(self.decoder_step1, rstar, lstar, tape, train)
        t = self._upsample_skip(self.decoder_step2, s, fstar, tape, train)
        u = self._upsample_skip(self.decoder_step3, t, bstar, tape, train)
        v = torch.empty(B, H, W, 64, **f32)
        if out_cells is None:
            self._conv(self.decoder_step4, u, img(v), 3, tape)
        else:
            m4 = self.decoder_step4
            if ops.h2_active() and _SPARSE_H2:
                # the fp16x2 product of the dense 3x3 layers (three 16-bit MFMAs per k step instead of fp32 MFMAs): the weights' planes
                # from the step's WeightPrep (or split per call), the input's bound = the max |u| its producer measured
                ui = img(u)
                w2, wa = ops._split_h2(ops.ohwi(m4.weight))
                call("df_sparse_conv3x3_h2", ptr(out_cells.key_sorted), ptr(out_cells.counts), B, ui, ptr(w2), ptr(ops.amax_of(ui, dev)),
                     ptr(wa), ptr(m4.bias.detach()), img(v), max(1, 256 // B), stream())
            elif ops.MFMA_BF16 and _SPARSE_H2:       # bf16-operand mode: one bf16 plane per operand, as that mode's dense convolutions
                call("df_sparse_conv3x3_bf16", ptr(out_cells.key_sorted), ptr(out_cells.counts), B, img(u), ptr(ops.ohwi(m4.weight)),
                     ptr(m4.bias.detach()), img(v), max(1, 256 // B), stream())
            else:
                call("df_sparse_conv3x3", ptr(out_cells.key_sorted), ptr(out_cells.counts), B, img(u), ptr(ops.ohwi(m4.weight)),
                     ptr(m4.bias.detach()), img(v), max(1, 256 // B), stream())
            if tape is not None:
                tape.append(("conv", m4, u, 3))
        return v

    def _conv(self, m: nn.Conv2d, x: torch.Tensor, y: DfImg, ks: int, tape: Optional[list], amax=None):
        ops.conv2d(img(x), ops.ohwi(m.weight), m.bias.detach(), y, ks, 1, amax_out=amax)
        if tape is not None:
            tape.append(("conv", m, x, ks))

    def _upsample_skip(self, m: UpsampleSkip, a: torch.Tensor, b: torch.Tensor, tape: Optional[list], train: bool = False) -> torch.Tensor:
        B, h, w, _ = a.shape
        lat, outc = m.u3.out_channels, m.u4_u5[1].out_channels
        dev = a.device
        f32 = dict(dtype=torch.float32, device=dev)
        # bf16-storage mode: the concatenation and the first 3x3 conv's output are bfloat16 (their consumers are the bf16-tile
        # 3x3 kernels); the block's output stays fp32 for the 1x1 conv / sparse kernels that read it
        s16 = bool(train and tape is not None and ops.BF16_STORE and ops.MFMA_BF16
                   and _stage_store16_ok(B, 2 * h, 2 * w, outc, dev) and _stage_store16_ok(B, 2 * h, 2 * w, 2 * lat, dev))
        mid = dict(dtype=torch.bfloat16 if s16 else torch.float32, device=dev)
        # pre-split mode (round 4): the concatenation and the first 3x3 conv's output are h2 images -- written split by the upsample
        # kernel / the 1x1 conv's epilogue / the 3x3 conv's epilogue, with scales from a-priori bounds (max |input| x the largest L1
        # norm of a weight row + max |bias|; the upsampled half: max |t|, measured) -- and read by LDS-DMA
        h2d = bool(train and tape is not None and not s16 and _h2p_ok(B, 2 * h, 2 * w, 2 * lat, outc, dev)
                   and _h2p_ok(B, 2 * h, 2 * w, outc, outc, dev))
        t = torch.empty(B, h, w, lat, **f32)
        # fp16x2 mode: ONE max |x| slot for the concatenation -- both 1x1 convolutions accumulate into it (the upsampled half is
        # made of convex combinations of t, so max |t| bounds it)
        cat_amax = ops.amax_slot(dev) if (ops.h2_active() and not s16) else None
        self._conv(m.u1_u2[0], a, img(t), 1, tape, amax=cat_amax)
        if h2d:
            w3 = ops.ohwi(m.u3.weight)
            cat_bound = ops.conv_out_bound(img(b), w3, m.u3.bias.detach(), dev, other=cat_amax)
            cat = ops.h2_empty((B, 2 * h, 2 * w, 2 * lat), dev, cat_bound)
        else:
            cat = torch.empty(B, 2 * h, 2 * w, 2 * lat, **mid)
        # round 6: where the concatenation is pre-split, the skip convolution's workgroups also write the bilinear half of their pixels
        # (one kernel writing whole pixels instead of two kernels one half each: ops.conv1x1_up_fused)
        fused = h2d and ops.conv1x1_up_fused(img(b), ops.ohwi(m.u3.weight), m.u3.bias.detach(), img(cat, lat, lat), img(t), self.align_corners)
        if not fused:
            ops.upsample2x(img(t), img(cat, lat, 0), self.align_corners)
        if tape is not None:
            tape.append(("up", h, w, lat))
        if fused:
            if tape is not None:
                tape.append(("conv", m.u3, b, 1))
        else:
            self._conv(m.u3, b, img(cat, lat, lat), 1, tape, amax=None if h2d else cat_amax)
        if cat_amax is not None and not h2d:
            cat._df_amax = (cat_amax, ver(cat))
        if h2d:
            u4 = ops.h2_empty((B, 2 * h, 2 * w, outc), dev,
                              ops.conv_out_bound(img(cat), ops.ohwi(m.u4_u5[0].weight), m.u4_u5[0].bias.detach(), dev))
        else:
            u4 = torch.empty(B, 2 * h, 2 * w, outc, **mid)
        self._conv(m.u4_u5[0], cat, img(u4), 3, tape)
        u5 = torch.empty(B, 2 * h, 2 * w, outc, **f32)
        self._conv(m.u4_u5[1], u4, img(u5), 3, tape)
        return u5

    # ------------------------------------------------------------------------------ bf16 inference ----
    def _bf16_weights(self):
        """bf16 copies of every conv weight ([O,kh,kw,I] memory; the first encoder conv zero-padded to 64 input
        channels), cached until a parameter changes (ops.PARAM_GEN / tensor versions)."""
        key = (ops.PARAM_GEN[0],) + tuple(p._version for p in self.parameters())
        c = getattr(self, "_df_bf16", None)
        if c is None or c[0] != key:
            wd = {}
            for name, m in self.named_modules():
                if isinstance(m, nn.Conv2d):
                    w = ops.ohwi(m.weight).detach()          # [O,kh,kw,I]
                    if w.shape[3] % 64:
                        w = torch.nn.functional.pad(w, (0, 64 - w.shape[3] % 64))
                    wd[m] = w.to(torch.bfloat16).contiguous()
            c = (key, wd)
            self._df_bf16 = c
        return c[1]

    def run_bf16(self, bstar: torch.Tensor) -> torch.Tensor:
        """Eval-mode forward with bf16 activations / weights on v_mfma_f32_32x32x16_bf16 (fp32 accumulation, fp32 folded
        BatchNorm + GELU epilogues): fp32 bstar [B,H,W,64] -> fp32 [B,H,W,64].  BASELINE configs[4] (inference)."""
        B, H, W, _ = bstar.shape
        dev = bstar.device
        bf = dict(dtype=torch.bfloat16, device=dev)
        wd = self._bf16_weights()
        s = stream()

        def dimg(t, c=None, c_off=0):   # descriptor of a bf16 NHWC tensor (element units)
            n, h, w, cc = t.shape
            c = cc - c_off if c is None else c
            return DfImg(t.data_ptr() + 2 * c_off, n, h, w, c, t.stride(2), n, t.stride(0), 0)

        def dpair(t, c):                # [B,h,w,2c] viewed as 2B images of c channels (cloud-major)
            n, h, w, _ = t.shape
            return DfImg(t.data_ptr(), 2 * n, h, w, c, t.stride(2), n, t.stride(0), c)

        def conv(m, x, y, ks, stride=1, bn=None, out_f32=0):
            if bn is None:
                call("df_conv2d_bf16", x, ptr(wd[m]), ptr(m.bias.detach()), y, ks, stride, ks // 2, ops.EPI_BIAS, None, None,
                     out_f32, s)
            else:
                scale, shift, _, _ = ops.folded_bn(bn)
                call("df_conv2d_bf16", x, ptr(wd[m]), ptr(m.bias.detach()), y, ks, stride, ks // 2, ops.EPI_BN_GELU, ptr(scale),
                     ptr(shift), out_f32, s)

        # network input: the two 32-channel canvases are the halves of one [B,H,W,64] tensor.  The first conv reads them
        # in place as 2B images whose 64-deep k chunk runs 32 channels past the cloud's own (into the other cloud / the
        # next pixel, zeros past the end of the buffer): its bf16 weights are zero-padded to 64 input channels, and the
        # canvas is finite, so those products are exact zeros -- no padded copy of the input.  (The last pixel of cloud 1
        # reaches 32 elements past the tensor: the allocation carries a zeroed tail so that those are zeros too.)
        buf16 = torch.empty(B * H * W * 64 + 64, **bf)
        buf16[-64:].zero_()
        bstar16 = buf16[:B * H * W * 64].view(B, H, W, 64)
        call("df_cast_bf16", ptr(bstar), ptr(bstar16), B * H * W, 64, 64, 64, s)
        x = DfImg(bstar16.data_ptr(), 2 * B, H, W, 64, 64, B, bstar16.stride(0), 32)
        cats, alive = [], [buf16]
        h, w = H, W
        for stage in (self.encoder_step_1, self.encoder_step_2, self.encoder_step_3):
            for i, m in enumerate(stage):
                if i == 0:
                    h, w = h // 2, w // 2
                C = m.conv.out_channels
                if i == len(stage) - 1:
                    keep = torch.empty(B, h, w, 2 * C, **bf)
                    cats.append(keep)
                    z = dpair(keep, C)
                else:
                    keep = torch.empty(2 * B, h, w, C, **bf)
                    z = dimg(keep)
                conv(m.conv, x, z, 3, m.stride, bn=m.batchnorm)
                alive.append(keep)
                x = z
        fstar, lstar, rstar = cats

        def upsample_skip(m, a, b):
            Bn, hh, ww, _ = a.shape
            lat, outc = m.u3.out_channels, m.u4_u5[1].out_channels
            t = torch.empty(Bn, hh, ww, lat, **bf)
            conv(m.u1_u2[0], dimg(a), dimg(t), 1)
            cat = torch.empty(Bn, 2 * hh, 2 * ww, 2 * lat, **bf)
            call("df_upsample2x_bf16", dimg(t), dimg(cat, lat, 0), int(self.align_corners), s)
            conv(m.u3, dimg(b), dimg(cat, lat, lat), 1)
            u4 = torch.empty(Bn, 2 * hh, 2 * ww, outc, **bf)
            conv(m.u4_u5[0], dimg(cat), dimg(u4), 3)
            u5 = torch.empty(Bn, 2 * hh, 2 * ww, outc, **bf)
            conv(m.u4_u5[1], dimg(u4), dimg(u5), 3)
            alive.extend([t, cat, u4])
            return u5

        sx = upsample_skip(self.decoder_step1, rstar, lstar)
        tx = upsample_skip(self.decoder_step2, sx, fstar)
        ux = upsample_skip(self.decoder_step3, tx, bstar16)
        v = torch.empty(B, H, W, 64, dtype=torch.float32, device=dev)
        conv(self.decoder_step4, dimg(ux), img(v), 3, out_f32=1)
        return v

    # ------------------------------------------------------------------------------ backward ----
    @staticmethod
    def _conv_bwd(m: nn.Conv2d, x: DfImg, dy: DfImg, ks: int, stride: int, dx: Optional[DfImg], acc_dx: bool,
                  grads: dict, with_bias: bool = True, wt: Optional[torch.Tensor] = None, bwd_bn=None) -> bool:
        """-> True if the data gradient's epilogue also produced the BatchNorm-backward partials asked for with bwd_bn"""
        w = ops.ohwi(m.weight)
        dev = w.device
        fused_bn = False
        if dx is not None:
            fused_bn = ops.conv2d(dy, ops.weight_transpose(w) if wt is None else wt, None, dx, ks, stride, mode=ops.CONV_DGRAD,
                                  accumulate=acc_dx, bwd_bn=bwd_bn)
        fused = with_bias and not _NO_FUSED_BIAS

        def wgrad():
            dw = torch.empty_like(w)  # [O,kh,kw,I] memory
            db = ops.conv2d_wgrad(x, dy, ks, stride, dw, want_bias=fused)
            grads[m.weight] = dw.permute(0, 3, 1, 2)  # logical [O,I,kh,kw], channels_last strides
            if with_bias:
                grads[m.bias] = db if fused else ops.colsum(dy, dev)

        if ops.SIDE is None:
            wgrad()
        else:
            with ops.SIDE.fork():
                wgrad()
        return fused_bn

    def run_backward(self, bstar: torch.Tensor, tape: list, dv: torch.Tensor, dbstar: Optional[torch.Tensor], grads: dict,
                     phase=None, sparse_input_grad: bool = False, dv_cells=None):
        """Consumes the tape of run(train=True).  dv [B,H,W,64].  Returns d(bstar) [B,H,W,64] (added to `dbstar`
        if given).  Parameter gradients go to `grads` {param: tensor}.
        sparse_input_grad: do NOT produce d(bstar); return (dy1, dskip) instead -- the output gradient of the first
        encoder conv [2B,H/2,W/2,64] and of the skip conv on bstar (a [B,H,W,64] channel slice) -- for a caller that only
        needs d(bstar) at occupied pillars (df_pillar_input_grad).
        dv_cells: PillarState of the cloud whose occupied cells are the ONLY non-zero pixels of dv (the decoder's gather
        backward wrote it): the last conv's weight gradient then sums over those cells only (df_sparse_wgrad3x3)."""
        B, H, W, _ = bstar.shape
        dev = bstar.device
        f32 = dict(dtype=torch.float32, device=dev)
        tape = list(tape)

        keep = ops.SIDE.keep if ops.SIDE is not None else None

        def pop(kind):
            e = tape.pop()
            assert e[0] == kind, (e[0], kind)
            if keep is not None:
                keep.append(e)
            return e

        def hold(t):
            if keep is not None:
                keep.append(t)
            return t

        def plain_conv_bwd(dy: torch.Tensor, dx: Optional[DfImg], acc: bool, wt=None):
            _, m, x, ks = pop("conv")
            self._conv_bwd(m, img(x), img(dy), ks, 1, dx, acc, grads, wt=wt)

        retained = {}

        def grad_like(x: torch.Tensor, dy: torch.Tensor, m: nn.Conv2d):
            """storage of d(x) for the data gradient of conv m (output gradient dy): pre-split when x was (its consumers are then the
            pre-split data- and weight-gradient kernels of the layer in front), with the a-priori bound max|dy| x max_row ||w^T row||_1;
            -> (tensor, transposed weights or None)"""
            if _is_h2(x):
                wt = ops.weight_transpose(ops.ohwi(m.weight))
                return hold(ops.h2_empty(x.shape, dev, ops.conv_out_bound(img(dy), wt, None, dev))), wt
            return hold(torch.empty_like(x)), None

        def upsample_skip_bwd(dout: torch.Tensor, a_like: Optional[torch.Tensor], db: Optional[DfImg], acc_b: bool):
            # reverse of: u1(a)->t ; up(t)->cat[:lat] ; u3(b)->cat[lat:] ; u4(cat) ; u5(u4)
            # (bf16-storage mode: u4 and cat were bfloat16 -> du4 is bfloat16 as well, so both 3x3 weight gradients see bf16
            #  x and dy; dcat stays fp32 for the 1x1 / upsample kernels behind it.  Pre-split mode: the same with h2 images.)
            # a_like: the tensor whose layout d(a) takes (the u4 of the block in front: bf16 / h2 / fp32), None = fp32
            # -> d(a) [the block's input gradient: the output gradient of the block in front]
            _, m5, x5, _ = tape[-1]
            du4, wt5 = grad_like(x5, dout, m5)
            plain_conv_bwd(dout, img(du4), False, wt5)
            _, m4, x4, _ = tape[-1]
            dcat = hold(torch.empty(x4.shape, **f32))
            plain_conv_bwd(du4, img(dcat), False)
            lat = dcat.shape[3] // 2
            # u3
            _, m3, xb, ks = pop("conv")
            self._conv_bwd(m3, img(xb), img(dcat, lat, lat), 1, 1, db, acc_b, grads)
            if db is None:
                retained["dskip"] = (dcat, lat)
            _, h, w, lat_ = pop("up")
            dt = hold(torch.empty(B, h, w, lat, **f32))
            dci = img(dcat, lat, 0)
            ops.upsample2x_bwd(dci, img(dt), self.align_corners)
            if getattr(dci, "_amax", None) is not None:
                # every input pixel of the bilinear x2 receives a total weight <= 2 per axis (4 in all; 4.5 covers align_corners
                # = True): max |dt| <= 4.5 max |dcat| -- a bound without a pass over dt, for the 1x1 data / weight gradients behind
                dt._df_amax = (ops.h2_bound(dci._amax, slack=4.5), ver(dt))
            _, m1, xa, ks = pop("conv")
            wt1 = None
            if a_like is not None and _is_h2(a_like):
                wt1 = ops.weight_transpose(ops.ohwi(m1.weight))
                dA = hold(ops.h2_empty(xa.shape, dev, ops.conv_out_bound(img(dt), wt1, None, dev)))
            else:
                dA = hold(torch.empty(xa.shape, dtype=torch.float32 if a_like is None else a_like.dtype, device=dev))
            self._conv_bwd(m1, img(xa), img(dt), 1, 1, img(dA), False, grads, wt=wt1)
            return dA

        # the gradient a block receives at its output (du, dT, dS) is bfloat16 when that block kept its u4 in bfloat16 (bf16-
        # storage mode): its two consumers are then the bf16-tile data- and weight-gradient kernels of the block's second conv
        def conv_x(k):      # input tensor of the k-th "conv" entry from the end of the tape
            convs = [e for e in tape if e[0] == "conv"]
            return convs[-k][2]
        # decoder_step4
        _, m, xu, _ = tape[-1]
        u4_3 = conv_x(2)                                           # conv entries from the end: step4, step3.u5 (x = u4)
        wt4 = ops.weight_transpose(ops.ohwi(m.weight))
        if _is_h2(u4_3):
            du = hold(ops.h2_empty(xu.shape, dev, ops.conv_out_bound(img(dv), wt4, None, dev)))
        else:
            du = hold(torch.empty(xu.shape, dtype=u4_3.dtype, device=dev))
        if dv_cells is None:
            plain_conv_bwd(dv, img(du), False, wt4)
        else:
            _, m, xu, _ = pop("conv")
            w4 = ops.ohwi(m.weight)
            ops.conv2d(img(dv), wt4, None, img(du), 3, 1, mode=ops.CONV_DGRAD)
            nblk = max(1, 256 // B)
            ws = torch.empty(nblk * B, 64 * 9 * 64, **f32)
            bws = torch.empty(nblk * B, 64, **f32)
            call("df_sparse_wgrad3x3_x2" if (_SPARSE_H2 and os.environ.get("DF_GRU_X2", "1") != "0") else "df_sparse_wgrad3x3", ptr(dv_cells.key_sorted), ptr(dv_cells.counts), B, img(xu), img(dv), ptr(ws), ptr(bws),
                 nblk, stream())
            dw4 = torch.empty_like(w4)
            db4 = torch.empty(64, **f32)
            call("df_conv2d_wgrad_reduce_bias", ptr(ws), nblk * B, 64, 9, 64, ptr(dw4), 9 * 64, 0, ptr(bws), ptr(db4), stream())
            grads[m.weight] = dw4.permute(0, 3, 1, 2)
            grads[m.bias] = db4
        # decoder_step3: a = T, b = bstar
        if sparse_input_grad:
            acc_b = False
        elif dbstar is None:
            dbstar = torch.empty(B, H, W, 64, **f32)
            acc_b = False
        else:
            acc_b = True
        # conv entries of a block in tape order: u1, u3, u4, u5 -> from the end: u5 (x = u4), u4, u3, u1
        dT = upsample_skip_bwd(du, conv_x(5), None if sparse_input_grad else img(dbstar), acc_b)   # step3 (4 entries), step2.u5 (x = u4)
        dF = hold(torch.empty(B, H // 2, W // 2, 128, **f32))   # d(fstar)
        dS = upsample_skip_bwd(dT, conv_x(5), img(dF), False)    # (tape shrank by one block) step1.u5
        dL = hold(torch.empty(B, H // 4, W // 4, 256, **f32))   # d(lstar)
        dR = upsample_skip_bwd(dS, None, img(dL), False)         # d(rstar) [B,H/8,W/8,512], fp32: the encoder's BatchNorm backward reads it
        if phase is not None:   # the four decoder steps are complete: their gradients can leave (optim.GradSink)
            phase([p for m in (self.decoder_step1, self.decoder_step2, self.decoder_step3, self.decoder_step4)
                   for p in m.parameters()])
        # encoder, stages 3..1; dz of a stage's last layer lives in the concatenated gradient buffer
        stage_in_grads = {3: (dL, 128), 2: (dF, 64), 1: (dbstar, 32)}
        dz = img_pair(dR, 256)
        pre_partial = None      # BatchNorm-backward partials of the layer about to be processed, left by the data gradient behind it
        for sidx, stage in ((3, self.encoder_step_3), (2, self.encoder_step_2), (1, self.encoder_step_1)):
            for i in reversed(range(len(stage))):
                pop("keep")
                _, m, x, y, bn_ss, ipg, groups, frozen = pop("cwn")
                # bf16-storage stage (y is bfloat16): dy of its stride-1 layers is bfloat16 too, and so is the dx they hand to
                # the layer in front; the stage's first (stride-2) layer keeps an fp32 dy for the fp32 kernels that consume it
                s16 = y.dtype == torch.bfloat16 and i > 0
                # pre-split stage: this layer's input x was an h2 image -> its dy is written as one too (both consumers -- the data
                # gradient into the layer in front and the weight gradient against x -- are the pre-split kernels)
                dy_h2 = x.elt == 2 and getattr(y, "_df_yamax", None) is not None
                dy, dgamma, dbeta, dbias = ops.bn_gelu_bwd(dz, y, bn_ss, ipg, groups, frozen=frozen,
                                                           dy_dtype=torch.bfloat16 if s16 else torch.float32, dy_h2=dy_h2,
                                                           y_amax=getattr(y, "_df_yamax", None) if dy_h2 else None,
                                                           partial_pre=pre_partial)
                pre_partial = None
                hold(dy)
                grads[m.batchnorm.weight], grads[m.batchnorm.bias], grads[m.conv.bias] = dgamma, dbeta, dbias
                bwd_bn = None
                if i > 0:
                    dxt = torch.empty(2 * B, x.h, x.w, x.c, dtype=torch.bfloat16 if s16 else torch.float32, device=dev)
                    dx, acc = img(dxt), False
                    if not s16 and ops.h2_active() and tape[-2][0] == "cwn":
                        # dz of the layer in front comes out of this layer's data gradient: let its epilogue also sum that layer's
                        # BatchNorm-backward partials (g, g xhat per tile and channel) -- one pass over dz and y less per layer
                        _, _, xq, yq, ssq, ipgq, groupsq, _ = tape[-2]
                        rows_pg = ipgq * x.h * x.w
                        tm = ops.conv_tile_m(rows_pg, x.c)
                        if rows_pg % tm == 0 and yq.dtype == torch.float32:
                            part = torch.empty(groupsq * (rows_pg // tm), x.c, 2, dtype=torch.float32, device=dev)
                            dx.grp_size = ipgq
                            dx.grp_off = ipgq * dxt.stride(0)
                            bwd_bn = (yq, ssq, part)
                            bwd_nbp = rows_pg // tm
                elif sidx == 1 and sparse_input_grad:
                    dx, acc = None, False
                    retained["dy1"] = dy
                else:
                    buf, c = stage_in_grads[sidx]
                    dx, acc = img_pair(buf, c), True
                if sidx == 1 and i == 0 and sparse_input_grad:
                    pass  # neither data nor weight gradient here: the caller evaluates both at occupied pillars only
                else:
                    fused = self._conv_bwd(m.conv, x, img(dy), 3, m.stride, dx, acc, grads, with_bias=False, bwd_bn=bwd_bn)
                    if fused:
                        pre_partial = (bwd_bn[2], bwd_nbp)
                dz = dx
            if phase is not None:
                phase(stage.parameters())
        assert not tape
        if sparse_input_grad:
            return retained["dy1"], retained["dskip"]
        return dbstar

    # -- reference-compatible call: backbone(pc0_img, pc1_img) on NCHW tensors (no autograd) -----------
    def forward(self, pc0_B: torch.Tensor, pc1_B: torch.Tensor) -> torch.Tensor:
        bstar = torch.cat([pc0_B, pc1_B], dim=1).permute(0, 2, 3, 1).contiguous()
        with torch.no_grad():
            v = self.run(bstar, self.training, None)
        return v.permute(0, 3, 1, 2)
