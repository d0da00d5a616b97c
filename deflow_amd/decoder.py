"""Point decoders with the reference's module trees ([REF decoder.py:72-199]): ConvGRU, ConvGRUDecoder,
LinearDecoder.  state_dict keys are identical to the reference classes (offset_encoder.*, gru.conv{z,r,q}.*,
decoder.{0,2}.*) so ``deflow_best.ckpt`` loads [REF deflow.py:41-47].  Compute: csrc/decoder.hip,
csrc/decoder_bwd.hip."""
from __future__ import annotations

import contextlib
import os

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import ops
from ._lib import DfGruWeights, DfGruWeightsT, DfImg, call, img, ptr, stream, ver


class ConvGRU(nn.Module):
    """Parameter container ([REF decoder.py:123-139]); the recurrence itself runs inside the fused decoder kernel."""

    def __init__(self, input_dim: int = 64, hidden_dim: int = 128):
        super().__init__()
        self.convz = nn.Conv1d(input_dim + hidden_dim, hidden_dim, 1)
        self.convr = nn.Conv1d(input_dim + hidden_dim, hidden_dim, 1)
        self.convq = nn.Conv1d(input_dim + hidden_dim, hidden_dim, 1)


@dataclass
class PointSet:
    """Padded per-sample decoder inputs (device-side counts: no host sync to launch)."""
    coords: torch.Tensor   # [B,N,3] i32 (z,y,x)
    offs: torch.Tensor     # [B,N,3] f32
    counts: torch.Tensor   # [B] i32
    # gather-backward helpers (pillar sort of pc0): original flat index sorted by cell, cell table, original->compact
    idx_sorted: Optional[torch.Tensor] = None
    cell_rng: Optional[torch.Tensor] = None
    cpos: Optional[torch.Tensor] = None


def pack_infos(infos: List[Dict[str, torch.Tensor]], H: int, W: int, device, need_bwd: bool) -> PointSet:
    """List-of-dicts (reference contract [REF decoder.py:185-199]) -> padded PointSet.  Host-side plumbing for the
    stand-alone head call; the fused DeFlow path never goes through here."""
    B = len(infos)
    ns = [int(i["voxel_coords"].shape[0]) for i in infos]
    N = max(1, max(ns))
    coords = torch.zeros(B, N, 3, dtype=torch.int32, device=device)
    offs = torch.zeros(B, N, 3, dtype=torch.float32, device=device)
    for b, i in enumerate(infos):
        coords[b, :ns[b]] = i["voxel_coords"].to(device=device, dtype=torch.int32)  # float coords accepted (.long() in ref)
        offs[b, :ns[b]] = i["point_offsets"].to(device=device, dtype=torch.float32)
    counts = torch.tensor(ns, dtype=torch.int32, device=device)
    ps = PointSet(coords, offs, counts)
    if need_bwd:
        ncells = B * H * W
        bidx = torch.arange(B, device=device, dtype=torch.int64)[:, None]
        key = bidx * (H * W) + coords[..., 1].long() * W + coords[..., 2].long()
        valid = torch.arange(N, device=device)[None, :] < counts[:, None]
        key = torch.where(valid, key, torch.full_like(key, ncells)).to(torch.int32).reshape(-1).contiguous()
        idx_sorted = torch.empty_like(key)
        cell_rng = torch.empty(ncells, 2, dtype=torch.int32, device=device)
        ws = torch.empty(call("df_cell_sort_ws_bytes", ncells), dtype=torch.uint8, device=device)
        # stable counting sort by cell (csrc/pillarize.hip): a cell's points become one run, ascending index inside it
        call("df_cell_sort", ptr(key), B * N, ncells, ptr(idx_sorted), ptr(cell_rng), ptr(ws), stream())
        ps.idx_sorted, ps.cell_rng = idx_sorted, cell_rng
        ps.cpos = torch.arange(N, device=device, dtype=torch.int32).repeat(B).contiguous()
    return ps


def _adjacent(a: torch.Tensor, b: torch.Tensor) -> bool:
    if not (a.is_contiguous() and b.is_contiguous() and a.data_ptr() + a.numel() * 4 == b.data_ptr()):
        return False
    sa, sb = a.untyped_storage(), b.untyped_storage()
    return sa.data_ptr() == sb.data_ptr()  # same arena: a strided view over both stays inside the storage


class ConvGRUDecoder(nn.Module):
    def __init__(self, pseudoimage_channels: int = 64, num_iters: int = 4):
        super().__init__()
        assert pseudoimage_channels == 64, "the fused HIP decoder is specialised for 64+64 channels"
        self.offset_encoder = nn.Linear(3, pseudoimage_channels)
        self.gru = ConvGRU(input_dim=pseudoimage_channels, hidden_dim=pseudoimage_channels * 2)
        self.decoder = nn.Sequential(nn.Linear(pseudoimage_channels * 3, pseudoimage_channels // 2), nn.GELU(),
                                     nn.Linear(pseudoimage_channels // 2, 3))
        self.num_iters = num_iters

    # -- weights as the kernels want them ------------------------------------------------------------
    def _weights(self) -> Tuple[DfGruWeights, list]:
        g = self.gru
        wz, wr = g.convz.weight.detach(), g.convr.weight.detach()
        bz, br = g.convz.bias.detach(), g.convr.bias.detach()
        # [z rows | r rows]: free when the parameter arena lays convz/convr out back to back, else one small cat -- cached
        # until a parameter changes (two launches per inference forward otherwise)
        if _adjacent(wz, wr) and _adjacent(bz, br):
            w_zr, b_zr = torch.as_strided(wz, (256, 192), (192, 1)), torch.as_strided(bz, (256,), (1,))
        else:
            key = (ops.PARAM_GEN[0], wz._version, wr._version, bz._version, br._version, wz.data_ptr(), wr.data_ptr())
            c = getattr(self, "_df_zr", None)
            if c is None or c[0] != key or torch.is_grad_enabled():
                c = (key, torch.cat([wz, wr], 0).view(256, 192), torch.cat([bz, br], 0))
                self._df_zr = c
            w_zr, b_zr = c[1], c[2]
        w_q = g.convq.weight.detach().view(128, 192)
        keep = [w_zr, b_zr, w_q]
        W = DfGruWeights(ptr(self.offset_encoder.weight.detach()), ptr(self.offset_encoder.bias.detach()), ptr(w_zr),
                         ptr(b_zr), ptr(w_q), ptr(g.convq.bias.detach()), ptr(self.decoder[0].weight.detach()),
                         ptr(self.decoder[0].bias.detach()), ptr(self.decoder[2].weight.detach()),
                         ptr(self.decoder[2].bias.detach()))
        return W, keep

    def _weights16(self, W: DfGruWeights, keep: list) -> Tuple[DfGruWeights, list]:
        """bf16 mode: the GEMM weights (w_zr, w_q, w_1) as bf16 copies -- one cast per use, i.e. per optimizer step -- for the
        kernels' mfma_bf16 = 2 form (bf16 weight tiles in LDS, csrc/gemm_dma.h); every other field stays fp32"""
        w_zr, _, w_q = keep
        c16 = [w_zr.to(torch.bfloat16), w_q.to(torch.bfloat16), self.decoder[0].weight.detach().to(torch.bfloat16)]
        W2 = DfGruWeights(W.w_off, W.b_off, ptr(c16[0]), W.b_zr, ptr(c16[1]), W.b_q, ptr(c16[2]), W.b_1, W.w_2, W.b_2)
        return W2, keep + c16

    @staticmethod
    def _x2_on() -> bool:
        """fp32 mode: the gate / head GEMMs of the forward and backward kernels as bf16x2 (two bf16 planes per operand, three
        MFMAs; csrc/gemm_dma.h WStreamT<3>) instead of fp32 MFMA.  DF_GRU_X2=0: the fp32-MFMA form."""
        return os.environ.get("DF_GRU_X2", "1") != "0" and not os.environ.get("DF_GRU_V1")

    @staticmethod
    def _lean_on() -> bool:
        """round 5: the lean decoder kernels (csrc/decoder4.hip).  DF_GRU_LEAN=0: the round-3/4 kernels (all planes saved), for A/B."""
        return os.environ.get("DF_GRU_LEAN", "1") != "0" and not os.environ.get("DF_GRU_V1")

    def _xtab(self, W: DfGruWeights) -> torch.Tensor:
        """[416,4] fp32: (W[:, 128:] W_off | W[:, 128:] b_off + b) of the z, r, q gates and the head's first layer -- the whole
        contribution of the offset encoding x = W_off o + b_off [REF decoder.py:172] as an affine map of the point's offsets.  One
        small launch per parameter state (per optimizer step in training; cached for inference)."""
        key = (ops.PARAM_GEN[0],) + tuple((p._version, p.data_ptr()) for p in self.parameters())
        c = getattr(self, "_df_xtab", None)
        if c is None or c[0] != key or torch.is_grad_enabled():
            t = torch.empty(416, 4, dtype=torch.float32, device=self.offset_encoder.weight.device)
            call("df_gru_xtab", W, ptr(t), stream())
            c = (key, t)
            self._df_xtab = c
        return c[1]

    @staticmethod
    def _split_x2(w: torch.Tensor) -> torch.Tensor:
        """[rows, cols] fp32 -> [rows, 2, cols] bfloat16 (hi | lo per row)"""
        w = w.contiguous()
        out = torch.empty(w.shape[0], 2, w.shape[1], dtype=torch.bfloat16, device=w.device)
        call("df_split_bf16x2_rows", ptr(w), ptr(out), w.shape[0], w.shape[1], stream())
        return out

    def _weights_x2(self, W: DfGruWeights, keep: list) -> Tuple[DfGruWeights, list]:
        w_zr, _, w_q = keep
        c = [self._split_x2(w_zr), self._split_x2(w_q), self._split_x2(self.decoder[0].weight.detach())]
        W2 = DfGruWeights(W.w_off, W.b_off, ptr(c[0]), W.b_zr, ptr(c[1]), W.b_q, ptr(c[2]), W.b_1, W.w_2, W.b_2)
        return W2, keep + c

    # -- engine ------------------------------------------------------------------------------------------
    def run(self, before: DfImg, after: DfImg, ps: PointSet, save: bool):
        """-> flow [B,N,3] (rows >= counts[b] are not written), save buffer or None."""
        B, N, _ = ps.coords.shape
        dev = ps.coords.device
        flow = torch.empty(B, N, 3, dtype=torch.float32, device=dev)
        T = self.num_iters
        sv = torch.empty((5 * T + 1) * B * N * 128, dtype=torch.float32, device=dev) if save else None
        W, keep = self._weights()
        bf = bool(ops.MFMA_BF16) and not os.environ.get("DF_GRU_V1")   # (the first-generation kernels are fp32 only)
        x2 = (not bf) and self._x2_on()
        # (the lean weight-gradient pass addresses a plane with 32-bit byte offsets: B * N * 512 < 2^31, ~4.19 M rows.  Beyond that a
        #  SAVING forward takes the round-4 kernels, whose backward has the generic fallback -- not a DF_E_SHAPE in the backward after
        #  the forward has kept only the lean planes)
        lean = self._lean_on() and not (save and B * N * 512 >= (1 << 31))
        xtab = self._xtab(W) if lean else None
        if bf:
            W, keep = self._weights16(W, keep)
        elif x2:
            W, keep = self._weights_x2(W, keep)
        if xtab is not None:
            # round 5 (csrc/decoder4.hip): x contributions from the [416,4] table, (T + 1) saved planes, gates recomputed backwards
            hs = torch.empty((T + 1) * B * N * 128, dtype=torch.float32, device=dev) if save else None
            with ops.timed("gru_fwd", flops=B * N * (589824.0 * T / 4 + 12870.0), bytes=B * N * (512.0 + 36.0), tag=f"T={T} save={save} lean",
                           moved=B * N * (512.0 * (1 + (T + 1 if save else 0)) + 36.0)):      # gather + (T + 1) hidden-state planes
                call("df_gru_lean_fwd", before, after, ptr(ps.coords), ptr(ps.offs), ptr(ps.counts), B, N, T, W, ptr(xtab), ptr(flow),
                     ptr(hs), 2 if bf else 3 if x2 else 0, stream())
            if hs is not None:
                hs.df_bf16, hs.df_lean, hs.df_xtab = bf, True, xtab
            return flow, hs
        # algorithmic work per point (un-hoisted count): 589 824 * T / 4 + 12 870 FLOP; fused-minimum traffic
        # 128 * 4 B gathered + 36 B of coordinates / offsets / flow
        with ops.timed("gru_fwd", flops=B * N * (589824.0 * T / 4 + 12870.0), bytes=B * N * (512.0 + 36.0), tag=f"T={T} save={save}",
                       moved=B * N * (512.0 * (1 + (5 * T + 1 if save else 0)) + 36.0)):        # gather + 5 T + 1 planes
            call("df_gru_decoder_fwd_mp", before, after, ptr(ps.coords), ptr(ps.offs), ptr(ps.counts), B, N, T, W, ptr(flow),
                 ptr(sv), 2 if bf else 3 if x2 else 0, stream())
        if sv is not None:
            # in bf16 mode planes 0..4 hold bf16 half rows (csrc/decoder3.hip): the backward kernels must run in the mode the
            # forward ran in, whatever ops.MFMA_BF16 says by then
            sv.df_bf16 = bf
        return flow, sv

    def run_bf16(self, before: DfImg, after: DfImg, ps: PointSet):
        """Inference forward with the gate GEMMs and the head's first layer on bf16 MFMA (fp32 state, gates and
        accumulation; BASELINE configs[4]).  bf16 weight copies are cached until a parameter changes."""
        B, N, _ = ps.coords.shape
        flow = torch.empty(B, N, 3, dtype=torch.float32, device=ps.coords.device)
        W, (w_zr, b_zr, w_q) = self._weights()
        key = (ops.PARAM_GEN[0],) + tuple(p._version for p in self.parameters())
        c = getattr(self, "_df_bf16", None)
        if c is None or c[0] != key:
            c = (key, w_zr.to(torch.bfloat16).contiguous(), w_q.to(torch.bfloat16).contiguous(),
                 self.decoder[0].weight.detach().to(torch.bfloat16).contiguous())
            self._df_bf16 = c
        _, wzr16, wq16, w116 = c
        g, d = self.gru, self.decoder
        call("df_gru_decoder_fwd_bf16", before, after, ptr(ps.coords), ptr(ps.offs), ptr(ps.counts), B, N, self.num_iters,
             ptr(self.offset_encoder.weight.detach()), ptr(self.offset_encoder.bias.detach()), ptr(wzr16), ptr(b_zr), ptr(wq16),
             ptr(g.convq.bias.detach()), ptr(w116), ptr(d[0].bias.detach()), ptr(d[2].weight.detach()), ptr(d[2].bias.detach()),
             ptr(flow), stream())
        return flow, None

    def run_backward(self, dflow: torch.Tensor, ps: PointSet, sv: torch.Tensor, dbefore: DfImg, dafter: DfImg,
                     acc_before: bool, acc_after: bool, grads: dict, before: DfImg = None, after: DfImg = None):
        if getattr(sv, "df_lean", False):
            return self._run_backward_lean(dflow, ps, sv, dbefore, dafter, acc_before, acc_after, grads)
        B, N, _ = ps.coords.shape
        dev, T, s = dflow.device, self.num_iters, stream()
        f32 = dict(dtype=torch.float32, device=dev)
        BN = B * N
        bf = int(getattr(sv, "df_bf16", ops.MFMA_BF16))   # the mode the planes were written in
        W, keep = self._weights()
        w_zr, b_zr, w_q = keep
        w1 = self.decoder[0].weight.detach()
        wt_zr = ops.weight_transpose(w_zr.view(256, 1, 1, 192)).view(192, 256)
        wt_q = ops.weight_transpose(w_q.view(128, 1, 1, 192)).view(192, 128)
        wt_1 = ops.weight_transpose(w1.view(32, 1, 1, 192)).view(192, 32)
        x2 = (not bf) and self._x2_on()
        if bf:   # the kernels' mfma_bf16 = 2 form: bf16 copies of the GEMM weights
            W, keep = self._weights16(W, keep)
            wt_zr, wt_q, wt_1 = wt_zr.to(torch.bfloat16), wt_q.to(torch.bfloat16), wt_1.to(torch.bfloat16)
        elif x2:   # mfma_bf16 = 3: two-plane rows of the GEMM weights (fp32 planes; the weight-gradient pass below is unchanged)
            W, keep = self._weights_x2(W, keep)
            wt_zr, wt_q, wt_1 = self._split_x2(wt_zr), self._split_x2(wt_q), self._split_x2(wt_1)
        WT = DfGruWeightsT(ptr(wt_zr), ptr(wt_q), ptr(wt_1))
        dh0, dx = torch.empty(BN, 128, **f32), torch.empty(BN, 64, **f32)
        dpre1, xbuf = torch.empty(BN, 32, **f32), torch.empty(BN, 64, **f32)
        dflow = dflow.contiguous()
        nblocks = B * ((N + 63) // 64)
        bias_partial = torch.zeros(nblocks, 772, **f32)
        # data gradients = the forward's GEMMs against the transposed weights: the same FLOP count (the weight gradients are
        # gru_wgrad's)
        with ops.timed("gru_bwd", flops=B * N * (589824.0 * T / 4 + 12870.0), bytes=B * N * (512.0 * 2 + 24.0),
                       moved=B * N * (512.0 * (4 * T + 1 + 3 * T + 1) + 24.0 + 256.0 + 384.0)):   # 4 T + 1 planes in, 3 T planes + dh0 (+ dx, x, dpre1) out
            call("df_gru_decoder_bwd_mp", ptr(dflow), ptr(ps.offs), ptr(ps.counts), B, N, T, W, WT, ptr(sv), ptr(dh0), ptr(dx),
                 ptr(dpre1), ptr(xbuf), ptr(bias_partial), 2 if bf else 3 if x2 else 0, s)
        bias_g = torch.empty(772, **f32)
        if nblocks >= 2048:  # tens of thousands of per-workgroup rows: two-stage column sum
            staged = torch.empty(64, 772, **f32)
            call("df_colsum_stage", ptr(bias_partial), nblocks, 772, 64, ptr(staged), s)
            call("df_colsum_finalize", ptr(staged), 64, 772, 1, ptr(bias_g), 0, s)
        else:
            call("df_colsum_finalize", ptr(bias_partial), nblocks, 772, 1, ptr(bias_g), 0, s)
        # image gradients: per-cell segmented sum (no atomics)
        ncell = dafter.h * dafter.w
        if dbefore is None:   # the caller evaluates d(before) sparsely from dh0 (df_pillar_input_grad)
            dbefore = DfImg(0, 0, 0, 0, 0, 0, 1, 0, 0)
        call("df_gather_bwd", ptr(dh0), ptr(ps.idx_sorted), ptr(ps.cell_rng), ptr(ps.cpos), B, N, dbefore, dafter,
             int(acc_before), int(acc_after), max(1, min(4096, ncell // 8)), s)
        side = ops.SIDE
        if side is not None:
            side.keep.extend([sv, xbuf, dpre1, ps])
        with (side.fork() if side is not None else contextlib.nullcontext()):
            s = stream()  # the side stream inside the fork
            # weight gradients: split-K GEMMs over the saved planes (now holding the gate pre-activation gradients)
            plane = T * BN * 128

            def rows_img(t: torch.Tensor, off: int, n_img: int, c: int, ld: int, img_stride: int) -> DfImg:
                return DfImg(t.data_ptr() + 4 * off, n_img, 1, BN, c, ld, n_img, img_stride, 0)

            h_in = rows_img(sv, 0 * plane, T, 128, 128, BN * 128)
            dz = rows_img(sv, 1 * plane, T, 128, 128, BN * 128)
            dr = rows_img(sv, 2 * plane, T, 128, 128, BN * 128)
            dq = rows_img(sv, 3 * plane, T, 128, 128, BN * 128)
            rh = rows_img(sv, 4 * plane, T, 128, 128, BN * 128)
            hT = rows_img(sv, 5 * plane, 1, 128, 128, BN * 128)
            x_rep = rows_img(xbuf, 0, T, 64, 64, 0)          # the same x rows for every iteration
            x_one = rows_img(xbuf, 0, 1, 64, 64, BN * 64)
            kw = dict(row_counts=ps.counts, rows_per_seg=N)
            if os.environ.get("DF_GRU_WGRAD_V1") and not bf:  # six generic 1x1 weight-gradient GEMMs (first generation), for A/B
                # (fp32 planes only: in bf16 mode the planes are bf16 half rows that only the fused kernel reads)
                dW_zr = torch.empty(256, 192, **f32)
                dW_q = torch.empty(128, 192, **f32)
                ops.conv2d_wgrad(h_in, dz, 1, 1, dW_zr, ld_co=192, dw_off=0, **kw)
                ops.conv2d_wgrad(x_rep, dz, 1, 1, dW_zr, ld_co=192, dw_off=128, **kw)
                ops.conv2d_wgrad(h_in, dr, 1, 1, dW_zr, ld_co=192, dw_off=128 * 192, **kw)
                ops.conv2d_wgrad(x_rep, dr, 1, 1, dW_zr, ld_co=192, dw_off=128 * 192 + 128, **kw)
                ops.conv2d_wgrad(rh, dq, 1, 1, dW_q, ld_co=192, dw_off=0, **kw)
                ops.conv2d_wgrad(x_rep, dq, 1, 1, dW_q, ld_co=192, dw_off=128, **kw)
            else:  # one fused streaming pass over the planes
                nsplit = call("df_gru_wgrad_splits")
                ws = torch.empty(nsplit, 384, 192, **f32)
                with ops.timed("gru_wgrad", flops=2.0 * 384 * 192 * B * N * T, bytes=3.0 * 128 * 192 * 4,
                               moved=B * N * T * 4.0 * (384 + 256 + 64)):     # 3 gate-gradient planes + h_in + r*h + x rows per step
                    call("df_gru_wgrad_mp", ptr(sv), ptr(xbuf), ptr(ps.counts), B, N, T, ptr(ws), nsplit, 3 if x2 else bf, s)
                dW_all = torch.empty(384, 192, **f32)
                call("df_conv2d_wgrad_reduce", ptr(ws), nsplit, 384, 1, 192, ptr(dW_all), 192, 0, s)
                dW_zr, dW_q = dW_all[:256], dW_all[256:]
            if x2 and os.environ.get("DF_GRU_HEAD_WGRAD", "1") != "0":
                # dW_1 [32,192] = dpre1^T [hT | x] in one streaming pass (round 4: gru_head_wgrad_kernel, bf16x2 products like the gate
                # kernels; was two generic 1x1 weight-gradient GEMMs on the transposed problem + a transpose: 0.83 ms per step)
                nsp1 = 128
                ws1 = torch.empty(nsp1, 32, 192, **f32)
                with ops.timed("gru_head_wgrad", flops=2.0 * 32 * 192 * B * N, bytes=B * N * 4.0 * (32 + 192)):
                    call("df_gru_head_wgrad", ptr(dpre1), sv.data_ptr() + 4 * 5 * plane, ptr(xbuf), ptr(ps.counts), B, N, ptr(ws1), nsp1, s)
                dW1 = torch.empty(32, 192, **f32)
                call("df_conv2d_wgrad_reduce", ptr(ws1), nsp1, 32, 1, 192, ptr(dW1), 192, 0, s)
            else:
                # dW_1^T [192,32] = [hT | x]^T dpre1  (output channels must be a multiple of 64 -> compute the transpose)
                dpre_img = rows_img(dpre1, 0, 1, 32, 32, BN * 32)
                dW1t = torch.empty(192, 32, **f32)
                with ops.mfma_bf16(bool(bf)):   # the head's generic weight gradients follow the forward's mode too
                    ops.conv2d_wgrad(dpre_img, hT, 1, 1, dW1t, ld_co=32, dw_off=0, **kw)
                    ops.conv2d_wgrad(dpre_img, x_one, 1, 1, dW1t, ld_co=32, dw_off=128 * 32, **kw)
                dW1 = ops.weight_transpose(dW1t.view(192, 1, 1, 32)).view(32, 192)
        g = self.gru
        grads[g.convz.weight] = dW_zr[:128].unsqueeze(2)
        grads[g.convr.weight] = dW_zr[128:].unsqueeze(2)
        grads[g.convq.weight] = dW_q.unsqueeze(2)
        grads[self.decoder[0].weight] = dW1
        grads[g.convz.bias] = bias_g[0:128]
        grads[g.convr.bias] = bias_g[128:256]
        grads[g.convq.bias] = bias_g[256:384]
        grads[self.decoder[0].bias] = bias_g[384:416]
        grads[self.offset_encoder.weight] = bias_g[416:608].view(64, 3)
        grads[self.offset_encoder.bias] = bias_g[608:672]
        grads[self.decoder[2].weight] = bias_g[672:768].view(3, 32)
        grads[self.decoder[2].bias] = bias_g[768:771]
        return dh0   # [B*N,128] gradient of the gathered rows (for a sparse d(before); see df_pillar_input_grad)

    def _run_backward_lean(self, dflow: torch.Tensor, ps: PointSet, hs: torch.Tensor, dbefore: DfImg, dafter: DfImg,
                           acc_before: bool, acc_after: bool, grads: dict):
        """backward of the lean decoder (csrc/decoder4.hip): hs = the forward's (T + 1) hidden-state planes; the data pass recomputes
        the gates and leaves dz_pre | dr_pre | dq_pre | r*h for the weight-gradient pass over the 128 h columns; every x-side gradient
        comes out of the [416,4] sums (df_gru_lean_finalize)."""
        B, N, _ = ps.coords.shape
        dev, T, s = dflow.device, self.num_iters, stream()
        f32 = dict(dtype=torch.float32, device=dev)
        BN = B * N
        bf = int(hs.df_bf16)
        W0, keep = self._weights()                       # the fp32 parameters (finalize reads them)
        w_zr, b_zr, w_q = keep
        w1 = self.decoder[0].weight.detach()
        wt_zr = ops.weight_transpose(w_zr.view(256, 1, 1, 192)).view(192, 256)
        wt_q = ops.weight_transpose(w_q.view(128, 1, 1, 192)).view(192, 128)
        wt_1 = ops.weight_transpose(w1.view(32, 1, 1, 192)).view(192, 32)
        x2 = (not bf) and self._x2_on()
        W = W0
        if bf:
            W, keep = self._weights16(W0, keep)
            wt_zr, wt_q, wt_1 = wt_zr.to(torch.bfloat16), wt_q.to(torch.bfloat16), wt_1.to(torch.bfloat16)
        elif x2:
            W, keep = self._weights_x2(W0, keep)
            wt_zr, wt_q, wt_1 = self._split_x2(wt_zr), self._split_x2(wt_q), self._split_x2(wt_1)
        WT = DfGruWeightsT(ptr(wt_zr), ptr(wt_q), ptr(wt_1))
        mode = 2 if bf else 3 if x2 else 0
        gpl = torch.empty(4 * T * BN * 128, **f32)
        dh0, dpre1 = torch.empty(BN, 128, **f32), torch.empty(BN, 32, **f32)
        dflow = dflow.contiguous()
        nblocks = B * ((N + 63) // 64)
        PW = call("df_gru_lean_partial_width")
        partial = torch.zeros(nblocks, PW, **f32)
        # data gradients: the forward's h-side GEMMs against the transposed weights + the recompute of the three gates
        with ops.timed("gru_bwd", flops=B * N * (589824.0 * T / 4 + 12870.0), bytes=B * N * (512.0 * 2 + 24.0),
                       moved=B * N * (512.0 * (T + 1 + 4 * T + 1) + 24.0 + 128.0)):    # T + 1 planes in, 4 T planes + dh0 (+ dpre1) out
            call("df_gru_lean_bwd", ptr(dflow), ptr(ps.offs), ptr(ps.counts), B, N, T, W, WT, ptr(hs.df_xtab), ptr(hs), ptr(gpl),
                 ptr(dh0), ptr(dpre1), ptr(partial), mode, s)
        sums = torch.empty(PW, **f32)
        if nblocks >= 2048:
            staged = torch.empty(64, PW, **f32)
            call("df_colsum_stage", ptr(partial), nblocks, PW, 64, ptr(staged), s)
            call("df_colsum_finalize", ptr(staged), 64, PW, 1, ptr(sums), 0, s)
        else:
            call("df_colsum_finalize", ptr(partial), nblocks, PW, 1, ptr(sums), 0, s)
        ncell = dafter.h * dafter.w
        if dbefore is None:   # the caller evaluates d(before) sparsely from dh0 (df_pillar_input_grad)
            dbefore = DfImg(0, 0, 0, 0, 0, 0, 1, 0, 0)
        src = getattr(dafter, "_src", None)
        if (src is not None and not acc_after and ops.h2_active() and os.environ.get("DF_GATHER_BWD_V1") is None
                and B * N < (1 << 29) and ncell < (1 << 28) and os.environ.get("DF_GATHER_AMAX", "1") != "0"):
            # the kernel measures max |d(after)| as it writes (round 5): the UNet backward's first data gradient asked for it with a
            # df_absmax pass over the whole image
            am = ops.amax_slot(dh0.device)
            call("df_gather_bwd_m", ptr(dh0), ptr(ps.idx_sorted), ptr(ps.cell_rng), ptr(ps.cpos), B, N, dbefore, dafter,
                 int(acc_before), max(1, min(4096, ncell // 8)), ptr(am), s)
            src._df_amax = (am, ver(src))
        else:
            call("df_gather_bwd", ptr(dh0), ptr(ps.idx_sorted), ptr(ps.cell_rng), ptr(ps.cpos), B, N, dbefore, dafter,
                 int(acc_before), int(acc_after), max(1, min(4096, ncell // 8)), s)
        side = ops.SIDE
        if side is not None:
            side.keep.extend([hs, gpl, dpre1, sums, ps])
        with (side.fork() if side is not None else contextlib.nullcontext()):
            s = stream()  # the side stream inside the fork
            nsplit = call("df_gru_wgrad_splits")
            ws = torch.empty(nsplit, 384, 128, **f32)
            with ops.timed("gru_wgrad", flops=2.0 * 384 * 192 * B * N * T, bytes=3.0 * 128 * 192 * 4,
                           moved=B * N * T * 4.0 * (384 + 256)):      # 3 gate-gradient planes + h_in + r*h per step
                call("df_gru_lean_wgrad", ptr(hs), ptr(gpl), ptr(ps.counts), B, N, T, ptr(ws), nsplit, 3 if x2 else (1 if bf else 0), s)
            dW_all = torch.empty(384, 192, **f32)
            call("df_conv2d_wgrad_reduce", ptr(ws), nsplit, 384, 1, 128, ptr(dW_all), 192, 0, s)
            # (round 5: 128 -> 1024 split-K workgroups -- 128 left half the CUs idle and 2.5 MB in flight: 0.42 ms for 0.8 GB of planes)
            nsp1 = int(os.environ.get("DF_GRU_HEAD_SPLITS", "1024"))
            ws1 = torch.empty(nsp1, 32, 128, **f32)
            with ops.timed("gru_head_wgrad", flops=2.0 * 32 * 192 * B * N, bytes=B * N * 4.0 * (32 + 192)):
                call("df_gru_lean_head_wgrad", ptr(dpre1), hs.data_ptr() + 4 * T * BN * 128, ptr(ps.counts), B, N, ptr(ws1), nsp1, s)
            dW1 = torch.empty(32, 192, **f32)
            call("df_conv2d_wgrad_reduce", ptr(ws1), nsp1, 32, 1, 128, ptr(dW1), 192, 0, s)
            dW_off, db_off, db = torch.empty(64, 3, **f32), torch.empty(64, **f32), torch.empty(416, **f32)
            call("df_gru_lean_finalize", ptr(sums), W0, ptr(dW_all), ptr(dW1), ptr(dW_off), ptr(db_off), ptr(db), s)
        g = self.gru
        grads[g.convz.weight] = dW_all[:128].unsqueeze(2)
        grads[g.convr.weight] = dW_all[128:256].unsqueeze(2)
        grads[g.convq.weight] = dW_all[256:].unsqueeze(2)
        grads[self.decoder[0].weight] = dW1
        grads[g.convz.bias] = db[0:128]
        grads[g.convr.bias] = db[128:256]
        grads[g.convq.bias] = db[256:384]
        grads[self.decoder[0].bias] = db[384:416]
        grads[self.offset_encoder.weight] = dW_off
        grads[self.offset_encoder.bias] = db_off
        grads[self.decoder[2].weight] = sums[1664:1760].view(3, 32)
        grads[self.decoder[2].bias] = sums[1760:1763]
        return dh0   # [B*N,128] gradient of the gathered rows (for a sparse d(before); see df_pillar_input_grad)

    # -- reference-compatible call ------------------------------------------------------------------------------
    def forward(self, before_pseudoimages: torch.Tensor, after_pseudoimages: torch.Tensor,
                voxelizer_infos: List[Dict[str, torch.Tensor]]) -> List[torch.Tensor]:
        """NCHW images + list of {"point_offsets", "voxel_coords"} -> list of [N_b,3] flows.  Differentiable w.r.t. the
        images and the parameters (autograd.Function over the HIP forward/backward)."""
        from .autograd import GruHeadFn
        H, W = before_pseudoimages.shape[2:]
        need_bwd = torch.is_grad_enabled() and (before_pseudoimages.requires_grad or after_pseudoimages.requires_grad or
                                                any(p.requires_grad for p in self.parameters()))
        ps = pack_infos(voxelizer_infos, H, W, before_pseudoimages.device, need_bwd)
        ns = [int(i["voxel_coords"].shape[0]) for i in voxelizer_infos]
        if need_bwd:
            flow = GruHeadFn.apply(self, ps, before_pseudoimages, after_pseudoimages, *self.parameters())
        else:
            bh = before_pseudoimages.permute(0, 2, 3, 1).contiguous()
            ah = after_pseudoimages.permute(0, 2, 3, 1).contiguous()
            flow, _ = self.run(img(bh), img(ah), ps, False)
        return [flow[b, :n] for b, n in enumerate(ns)]


class LinearDecoder(nn.Module):
    def __init__(self, pseudoimage_channels: int = 64):
        super().__init__()
        assert pseudoimage_channels == 64
        self.offset_encoder = nn.Linear(3, 128)
        self.decoder = nn.Sequential(nn.Linear(pseudoimage_channels * 4, 32), nn.GELU(), nn.Linear(32, 3))

    def run(self, before: DfImg, after: DfImg, ps: PointSet, save: bool = False):
        """-> flow [B,N,3], None (the backward recomputes the gather and the hidden layer instead of saving them)"""
        B, N, _ = ps.coords.shape
        flow = torch.empty(B, N, 3, dtype=torch.float32, device=ps.coords.device)
        d = self.decoder
        call("df_linear_decoder_fwd", before, after, ptr(ps.coords), ptr(ps.offs), ptr(ps.counts), B, N,
             ptr(self.offset_encoder.weight.detach()), ptr(self.offset_encoder.bias.detach()), ptr(d[0].weight.detach()),
             ptr(d[0].bias.detach()), ptr(d[2].weight.detach()), ptr(d[2].bias.detach()), ptr(flow), stream())
        return flow, None

    def run_backward(self, dflow: torch.Tensor, ps: PointSet, sv, dbefore: DfImg, dafter: DfImg, acc_before: bool,
                     acc_after: bool, grads: dict, before: DfImg = None, after: DfImg = None):
        B, N, _ = ps.coords.shape
        dev, s = dflow.device, stream()
        f32 = dict(dtype=torch.float32, device=dev)
        BN = B * N
        d = self.decoder
        w1 = d[0].weight.detach()
        wt_1 = ops.weight_transpose(w1.view(32, 1, 1, 256)).view(256, 32)
        vx, dh0, dxe = torch.empty(BN, 256, **f32), torch.empty(BN, 128, **f32), torch.empty(BN, 128, **f32)
        dpre1, hid = torch.empty(BN, 32, **f32), torch.empty(BN, 32, **f32)
        dflow = dflow.contiguous()
        call("df_linear_decoder_bwd", before, after, ptr(ps.coords), ptr(ps.offs), ptr(ps.counts), ptr(dflow), B, N,
             ptr(self.offset_encoder.weight.detach()), ptr(self.offset_encoder.bias.detach()), ptr(w1),
             ptr(d[0].bias.detach()), ptr(d[2].weight.detach()), ptr(wt_1), ptr(vx), ptr(dh0), ptr(dxe), ptr(dpre1),
             ptr(hid), s)
        ncell = dafter.h * dafter.w
        if dbefore is None:
            dbefore = DfImg(0, 0, 0, 0, 0, 0, 1, 0, 0)
        call("df_gather_bwd", ptr(dh0), ptr(ps.idx_sorted), ptr(ps.cell_rng), ptr(ps.cpos), B, N, dbefore, dafter,
             int(acc_before), int(acc_after), max(1, min(4096, ncell // 8)), s)
        # dW1^T [256,32] = vx^T dpre1 (row GEMM with the validity mask), then transpose back
        rows = lambda t, c: DfImg(t.data_ptr(), 1, 1, BN, c, c, 1, BN * c, 0)
        dW1t = torch.empty(256, 32, **f32)
        ops.conv2d_wgrad(rows(dpre1, 32), rows(vx, 256), 1, 1, dW1t, ld_co=32, row_counts=ps.counts, rows_per_seg=N)
        grads[d[0].weight] = ops.weight_transpose(dW1t.view(256, 1, 1, 32)).view(32, 256)
        so = lambda a, lda, na, b, ldb, nb: ops.small_outer(a, lda, na, b, ldb, nb, ps.counts, N, B, BN)
        grads[d[0].bias] = so(dpre1, 32, 32, None, 0, 1).view(32)
        dfl = dflow.view(BN, 3)
        grads[d[2].weight] = so(dfl, 3, 3, hid, 32, 32)
        grads[d[2].bias] = so(dfl, 3, 3, None, 0, 1).view(3)
        offs = ps.offs.view(BN, 3)
        grads[self.offset_encoder.weight] = torch.cat([so(dxe, 128, 64, offs, 3, 3), so(dxe[:, 64:], 128, 64, offs, 3, 3)], 0)
        grads[self.offset_encoder.bias] = so(dxe, 128, 128, None, 0, 1).view(128)
        return dh0

    def forward(self, before_pseudoimages, after_pseudoimages, voxelizer_infos):
        from .autograd import GruHeadFn
        H, W = before_pseudoimages.shape[2:]
        need_bwd = torch.is_grad_enabled() and (before_pseudoimages.requires_grad or after_pseudoimages.requires_grad or
                                                any(p.requires_grad for p in self.parameters()))
        ps = pack_infos(voxelizer_infos, H, W, before_pseudoimages.device, need_bwd)
        ns = [int(i["voxel_coords"].shape[0]) for i in voxelizer_infos]
        if need_bwd:
            flow = GruHeadFn.apply(self, ps, before_pseudoimages, after_pseudoimages, *self.parameters())
        else:
            bh = before_pseudoimages.detach().permute(0, 2, 3, 1).contiguous()
            ah = after_pseudoimages.detach().permute(0, 2, 3, 1).contiguous()
            with torch.no_grad():
                flow, _ = self.run(img(bh), img(ah), ps, False)
        return [flow[b, :n] for b, n in enumerate(ns)]
