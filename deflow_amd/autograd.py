"""torch.autograd plumbing around the HIP engines.  PyTorch only chains these nodes and owns the memory; every
forward and backward computation inside them is a HIP kernel launched through the C ABI."""
from __future__ import annotations

import os
from typing import List

import torch

from . import ops
from ._lib import img, call, ptr, stream
from .unet import FastFlow3DUNet


class GradDict(dict):
    """parameter -> gradient, keyed by storage address (robust to autograd re-wrapping tensor objects)"""

    def __setitem__(self, p: torch.Tensor, g: torch.Tensor):
        super().__setitem__(p.data_ptr(), g)

    def lookup(self, p: torch.Tensor):
        return super().get(p.data_ptr())


def _grads_for(params: List[torch.Tensor], grads: "GradDict", sink=None):
    out = []
    for p in params:
        if sink is not None and sink.was_delivered(p):   # already copied into the gradient arena (optim.GradSink)
            out.append(None)
            continue
        g = grads.lookup(p)
        if g is not None and g.shape != p.shape:
            g = g.reshape(p.shape)
        out.append(g)
    return tuple(out)


class GruHeadFn(torch.autograd.Function):
    """Stand-alone differentiable ConvGRUDecoder call on NCHW images (used by the reference-compatible head API)."""

    @staticmethod
    def forward(ctx, head, ps, before, after, *params):
        bh = before.detach().permute(0, 2, 3, 1).contiguous()
        ah = after.detach().permute(0, 2, 3, 1).contiguous()
        flow, sv = head.run(img(bh), img(ah), ps, True)
        ctx.head, ctx.ps, ctx.sv, ctx.shape, ctx.params = head, ps, sv, bh.shape, list(params)
        ctx.imgs = (bh, ah)
        return flow

    @staticmethod
    def backward(ctx, dflow):
        head, ps = ctx.head, ctx.ps
        B, H, W, _ = ctx.shape
        dev = dflow.device
        db = torch.empty(B, H, W, 64, dtype=torch.float32, device=dev)
        da = torch.empty(B, H, W, 64, dtype=torch.float32, device=dev)
        grads = GradDict()
        head.run_backward(dflow, ps, ctx.sv, img(db), img(da), False, False, grads, before=img(ctx.imgs[0]),
                          after=img(ctx.imgs[1]))
        if ops.SIDE is not None:
            ops.SIDE.join()
        ctx.sv = ctx.imgs = None
        return (None, None, db.permute(0, 3, 1, 2), da.permute(0, 3, 1, 2)) + _grads_for(ctx.params, grads)


class DeFlowFn(torch.autograd.Function):
    """The whole hot path as one autograd node: pillarise both clouds -> UNet -> decoder, with a hand-sequenced
    backward (decoder -> UNet -> pillar feature net).  Inputs: the ego-compensated pc0 and pc1 (no gradient) and
    every parameter; output: padded flow [B,N,3]."""

    @staticmethod
    def forward(ctx, model, pc0s, pc1s, *params):
        flow, state = model._run(pc0s, pc1s, train=model.training, save=True)   # eval mode: frozen-BatchNorm tape
        ctx.model, ctx.state, ctx.params = model, state, list(params)
        model._state_tmp = state
        return flow

    @staticmethod
    def backward(ctx, dflow):
        sink = getattr(ctx.model, "_grad_sink", None)
        grads = deflow_backward(ctx.model, ctx.state, dflow, ctx.params, sink)
        ctx.state = None
        return (None, None, None) + _grads_for(ctx.params, grads, sink)


TAP = None   # debugging / tests: callable(stage_name, **tensors) called at the hand-over points of the backward


def deflow_backward(model, st: dict, dflow: torch.Tensor, params: List[torch.Tensor], sink=None) -> "GradDict":
    """The hand-sequenced backward of the whole hot path (decoder -> UNet -> pillar feature net) on the state `_run(save=True)`
    left: every launch is a HIP kernel through the C ABI.  Called by DeFlowFn.backward (autograd users) and DIRECTLY by
    optim.Trainer.step (no autograd engine, no worker thread: the launch sequence stays on the caller's thread, which is what
    lets the data-parallel step be captured as HIP-graph segments split at the gradient buckets).
    Gradients leave through `sink` phase by phase (optim.GradSink); what the sink did not take is in the returned dict."""
    grads = GradDict()
    bstar = st["bstar"]
    if ops.SIDE is None and os.environ.get("DF_SIDE_STREAM") == "1":
        ops.SIDE = ops.SideStream(bstar.device)
    # phase callback: hand finished gradients to the arena / the overlapped all-reduce.  With weight gradients in flight on
    # the side stream a phase first joins it -- only worth it when a collective is waiting for the bucket (data-parallel
    # ranks); on one rank everything is delivered once at the end
    if sink is None or (ops.SIDE is not None and not sink.collective):
        phase = lambda ps: None
    elif ops.SIDE is not None:
        def phase(ps):
            ops.SIDE.join()
            sink.deliver(ps, grads)
    else:
        phase = lambda ps: sink.deliver(ps, grads)
    B, H, W, _ = bstar.shape
    dev = bstar.device
    dbstar = torch.empty(B, H, W, 64, dtype=torch.float32, device=dev)
    dv = torch.empty(B, H, W, 64, dtype=torch.float32, device=dev)
    # d(bstar) is only read at occupied pillars (pillar feature net backward): the UNet's two data gradients into it
    # -- skip conv and first encoder conv -- are evaluated there only (df_pillar_input_grad) instead of densely for
    # all H*W cells.  DF_DENSE_CANVAS_GRAD=1 keeps the dense kernels (A/B, tests).
    sparse = os.environ.get("DF_DENSE_CANVAS_GRAD") != "1" and isinstance(model.backbone, FastFlow3DUNet)
    if sparse:
        # decoder: its gather backward writes d(before) = d(bstar) and d(after) = dv densely (a cheap stream)
        dh0 = model.head.run_backward(dflow, st["ps"], st["sv"], img(dbstar), img(dv), False, False, grads,
                                      before=img(bstar), after=img(st["v"]))
        if TAP is not None:
            TAP("head", dh0=dh0, dbstar=dbstar, dv=dv)
        del dh0
        st["sv"] = None
        phase(list(model.head.parameters()))
        dy1, (dcat, lat) = model.backbone.run_backward(bstar, st["tape"], dv, None, grads, phase, sparse_input_grad=True,
                                                       dv_cells=st["p0"])
        st["tape"] = None
        bb = model.backbone
        w1 = ops.ohwi(bb.encoder_step_1[0].conv.weight)
        w3 = ops.ohwi(bb.decoder_step3.u3.weight)
        nb = max(1, 256 // B)
        dw1 = torch.empty_like(w1)   # [64,3,3,32] memory
        for cloud, pst in ((0, st["p0"]), (1, st["p1"])):
            ws1 = torch.empty(nb * B, 64 * 9 * 32, dtype=torch.float32, device=dev)
            call("df_sparse_in_wgrad", ptr(pst.key_sorted), ptr(pst.counts), B, H, W, cloud, ptr(dy1),
                 img(bstar, 32, 32 * cloud), ptr(ws1), nb, stream())
            call("df_conv2d_wgrad_reduce", ptr(ws1), nb * B, 64, 9, 32, ptr(dw1), 9 * 32, cloud, stream())
            call("df_pillar_input_grad", ptr(pst.key_sorted), ptr(pst.counts), B, H, W, cloud, ptr(dy1), ptr(w1),
                 img(dcat, lat, lat), ptr(w3), img(dbstar, 32, 32 * cloud), 1, nb, stream())  # one 16-wave workgroup per CU
        ops.wrote(dbstar)
        grads[bb.encoder_step_1[0].conv.weight] = dw1.permute(0, 3, 1, 2)
        if TAP is not None:
            TAP("canvas_grad", dy1=dy1, dskip=dcat[..., lat:], dbstar=dbstar)
    else:
        # decoder: writes d(before)=d(bstar) and d(after)=dv completely (zeros where no point looked)
        model.head.run_backward(dflow, st["ps"], st["sv"], img(dbstar), img(dv), False, False, grads,
                                before=img(bstar), after=img(st["v"]))
        st["sv"] = None
        phase(list(model.head.parameters()))
        # UNet: accumulates its own d(bstar) into the same buffer
        model.backbone.run_backward(bstar, st["tape"], dv, dbstar, grads, phase)
        st["tape"] = None
    # pillar feature net of both clouds (shared weights -> accumulate)
    emb = model.embedder
    g = emb.pillarize_bwd(st["p0"], img(dbstar, 32, 0), None)
    g = emb.pillarize_bwd(st["p1"], img(dbstar, 32, 32), g)
    grads[emb._lin.weight], grads[emb._bn.weight], grads[emb._bn.bias] = g
    if TAP is not None:
        TAP("pfn", dW=g[0], dgamma=g[1], dbeta=g[2], dbstar=dbstar, **{f"{k}{i}": getattr(st[f"p{i}"], k) for i in (0, 1)
                                                                      for k in ("bn_ss", "pts_sorted", "key_sorted", "cell_rng", "counts")})
    if ops.SIDE is not None:
        ops.SIDE.join()
    if sink is not None:
        sink.deliver(params, grads)   # the pillar feature net, and everything else if phases were off
    return grads


class DeflowLossFn(torch.autograd.Function):
    """deflowLoss summed over the batch on the padded flow tensor (rows < counts[b])."""

    @staticmethod
    def forward(ctx, est, gt, counts):
        B, N, _ = est.shape
        dev = est.device
        nblk = max(1, min(32, (N + 255) // 256))
        partial = torch.empty(B, nblk, 6, dtype=torch.float32, device=dev)
        est_c, gt_c = est.contiguous(), gt.contiguous()
        call("df_deflow_loss_fwd", ptr(est_c), ptr(gt_c), ptr(counts), B, N, ptr(partial), nblk, stream())
        bins = torch.empty(B, 6, dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        call("df_deflow_loss_finalize", ptr(partial), B, nblk, ptr(bins), ptr(loss), stream())
        ctx.save_for_backward(est_c, gt_c, counts, bins)
        ctx.nblk = nblk
        return loss[0]

    @staticmethod
    def backward(ctx, gloss):
        est, gt, counts, bins = ctx.saved_tensors
        B, N, _ = est.shape
        dest = torch.empty_like(est)
        g = gloss.reshape(1).contiguous().float()
        call("df_deflow_loss_bwd", ptr(est), ptr(gt), ptr(counts), B, N, ptr(bins), ptr(g), 1.0, ptr(dest), ctx.nblk,
             stream())
        return dest, None, None
