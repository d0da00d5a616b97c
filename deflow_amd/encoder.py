"""DynamicEmbedder: points -> 32-channel BEV pseudo-image + per-point voxel info.

Mirrors the reference's constructor and return contract ([REF deflow.py:27-30,82-83,97-101]); the state_dict
keys follow the upstream module tree (``feature_net.pfn_layers.0.{0,1}.*``).  All compute is HIP
(csrc/pillarize.hip): voxelise -> stable compaction -> stable radix sort by cell -> per-cell gather-reduce.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import ops
from ._lib import DfGeom, DfImg, call, img, ptr, stream


def canvas_alloc(*shape, device) -> torch.Tensor:
    """canvas buffer for pillarize(): the band pipeline writes every byte itself (zeros included), so no fill here"""
    return torch.empty(*shape, dtype=torch.float32, device=device)


def make_geom(voxel_size, point_cloud_range) -> DfGeom:
    vs = torch.tensor(voxel_size, dtype=torch.float32)
    rg = torch.tensor(point_cloud_range, dtype=torch.float32)
    grid = torch.round((rg[3:] - rg[:3]) / vs).long()  # mmcv Voxelization.__init__
    off = [float(voxel_size[i]) / 2 + float(point_cloud_range[i]) for i in range(3)]  # python doubles, as upstream
    return DfGeom(float(vs[0]), float(vs[1]), float(vs[2]), float(rg[0]), float(rg[1]), float(rg[2]),
                  off[0], off[1], off[2], int(grid[0]), int(grid[1]), int(grid[2]))


@dataclass
class PillarState:
    """Device-side result of pillarising one cloud set [B,N,3]; padded, no host sync needed to use it."""
    pts: torch.Tensor          # [B,N,3] the (ego-compensated) input
    counts: torch.Tensor       # [B] i32 valid points per sample
    points_c: torch.Tensor     # [B,N,3] compacted
    coords_c: torch.Tensor     # [B,N,3] i32 (z,y,x)
    idx_c: torch.Tensor        # [B,N] i64 index into the padded input
    offs_c: torch.Tensor       # [B,N,3]
    cpos: torch.Tensor         # [B*N] i32 original -> compact
    idx_sorted: torch.Tensor   # [B*N] u32 (as i32) original flat index, sorted by cell
    cell_rng: torch.Tensor     # [B*H*W,2] i32
    key_sorted: torch.Tensor   # [B*N] u32 (as i32) sorted cell keys
    pts_sorted: torch.Tensor   # [B*N,3] points gathered into sorted order
    bn_ss: torch.Tensor        # [B or 1,4,32]
    bn_stride: int             # 128 (per-sample stats) or 0

    def split(self, B: int):
        """state of a merged [2B,N,3] set -> (first B samples, last B samples).  Per-sample arrays are sliced; the
        globally sorted arrays (idx_sorted / key_sorted / pts_sorted: the first set's segment comes first, the second
        starts at a device-side offset) stay whole in the first half and are dropped from the second -- only the
        tape-less forward uses merged sets, and it reads them for the first cloud alone."""
        N = self.pts.shape[1]
        nc = 0 if self.cell_rng is None else self.cell_rng.shape[0] // 2
        per = self.bn_stride != 0
        cr = self.cell_rng
        a = PillarState(self.pts[:B], self.counts[:B], self.points_c[:B], self.coords_c[:B], self.idx_c[:B], self.offs_c[:B],
                        self.cpos[:B * N], self.idx_sorted, None if cr is None else cr[:nc], self.key_sorted, self.pts_sorted,
                        self.bn_ss[:B] if per else self.bn_ss, self.bn_stride)
        b = PillarState(self.pts[B:], self.counts[B:], self.points_c[B:], self.coords_c[B:], self.idx_c[B:], self.offs_c[B:],
                        self.cpos[B * N:], None, None if cr is None else cr[nc:], None, None,
                        self.bn_ss[B:] if per else self.bn_ss, self.bn_stride)
        return a, b


class _FeatureNet(nn.Module):
    """Parameter container with the upstream names (DynamicPillarFeatureNet, feat_channels=(32,), mode='avg')."""

    def __init__(self, feat_channels: int):
        super().__init__()
        self.pfn_layers = nn.ModuleList([
            nn.Sequential(nn.Linear(9, feat_channels, bias=False),
                          nn.BatchNorm1d(feat_channels, eps=1e-3, momentum=0.01), nn.ReLU(inplace=True))])


class DynamicEmbedder(nn.Module):
    def __init__(self, voxel_size, pseudo_image_dims, point_cloud_range, feat_channels: int = 32, mode: str = "avg"):
        super().__init__()
        assert feat_channels == 32, "the HIP pillar feature net is specialised for 32 channels"
        self.voxel_size = list(voxel_size)
        self.point_cloud_range = list(point_cloud_range)
        self.geom = make_geom(voxel_size, point_cloud_range)
        self.H, self.W = int(pseudo_image_dims[0]), int(pseudo_image_dims[1])
        assert (self.geom.gy, self.geom.gx) == (self.H, self.W), "pseudo_image_dims must equal the voxel grid"
        self.mode = 0 if mode == "avg" else 1
        self.feature_net = _FeatureNet(feat_channels)

    # -- parameters --------------------------------------------------------------------------
    @property
    def _lin(self) -> nn.Linear:
        return self.feature_net.pfn_layers[0][0]

    @property
    def _bn(self) -> nn.BatchNorm1d:
        return self.feature_net.pfn_layers[0][1]

    # -- engine ------------------------------------------------------------------------------
    def pillarize(self, pts: torch.Tensor, out: DfImg, train: bool, need_cells: bool = True, occ: Optional[torch.Tensor] = None,
                  amax: Optional[torch.Tensor] = None) -> PillarState:
        """pts [B,N,3] f32 contiguous on the GPU; writes the WHOLE [B,H,W,32] canvas `out` (zeros included: the canvas needs no
        prior fill).  need_cells: also leave the dense per-cell [start, end) table the backward kernels read.
        occ (round 5): the PERSISTENT-canvas form -- `out` is a buffer this embedder wrote last time with the same `occ`
        ([S][bands][64] int32 occupancy words from `occ_alloc`, zero at first together with the buffer): only occupied cells and the
        cells occupied last time are written (df_pillar2_band_sp); the result is the same canvas.  amax (with occ; a zeroed device scalar,
        shared by the calls that fill one canvas): receives the maximum canvas value written."""
        B, N, _ = pts.shape
        # algorithmic traffic of the stage: the points once in, the dense 32-channel canvas once out
        with ops.timed("pillarise_fwd", bytes=B * (N * 12.0 + 32.0 * self.H * self.W * 4.0), tag=f"B={B} N={N}"):
            return self._pillarize(pts, out, train, need_cells, occ, amax)

    def bands(self, S: int) -> Tuple[int, int]:
        """(rows per band, bands per sample) of the band pipeline for a set of S samples"""
        H, W = self.H, self.W
        R = call("df_pillar2_rows_per_band", H, W)
        if R <= 0:
            raise RuntimeError(f"pillarise: a {H}x{W} grid is not supported (rows wider than 2048 cells)")
        # few samples (B = 1 inference: S = 2): thinner bands, so that the band kernel still has >= ~2048 workgroups to spread
        # over the 256 CUs (one workgroup per (band, sample); its latency chain shortens with the band)
        rmin = int(os.environ.get("DF_P2_MIN_WGS", "2048"))
        while R > 1 and S * ((H + R - 1) // R) < rmin and (H + R // 2 - 1) // (R // 2) <= 512:
            R //= 2
        return R, (H + R - 1) // R

    def occ_alloc(self, S: int, device) -> torch.Tensor:
        """zeroed occupancy words of a persistent canvas for sets of S samples"""
        return torch.zeros(S, self.bands(S)[1], 64, dtype=torch.int32, device=device)

    def _bn_state(self, train: bool, partial, counts, B: int, nbs: int, dev):
        """-> (bn_ss, bn_stride): per-sample batch statistics (training; running statistics updated) or the folded running ones"""
        bn, s = self._bn, stream()
        if train:
            if ops.SYNC is not None:
                bn_ss = self._sync_bn_stats(partial, counts)
            else:
                bn_ss = torch.empty(B, 4, 32, dtype=torch.float32, device=dev)
                # fp16x2 mode: the finalisation also leaves an a-priori bound of max |canvas| (self.canvas_bound: one slot shared by
                # the clouds of a forward, handed to the canvas tensor by DeFlow._run) -- no pass over the 1 GB canvas for it
                cb = getattr(self, "canvas_bound", None)
                call("df_pfn_bn_finalize2", ptr(partial), B, nbs, ptr(counts), ptr(bn.weight.detach()), ptr(bn.bias.detach()),
                     bn.eps, bn.momentum, ptr(bn.running_mean), ptr(bn.running_var), ptr(bn_ss),
                     ptr(self._lin.weight.detach()) if cb is not None else None, self.geom, ptr(cb), s)
            ops.bump_tracked(bn.num_batches_tracked, B)
            ops.PARAM_GEN[0] += 1
            return bn_ss, 128
        c = getattr(bn, "_df_fold_ss", None)
        fold = ops.folded_bn(bn)
        if c is None or c[0] is not fold[0]:   # re-stack only when the fold was recomputed
            c = (fold[0], torch.stack(list(fold)).contiguous())
            bn._df_fold_ss = c
        return c[1], 0

    def _pillarize(self, pts: torch.Tensor, out: DfImg, train: bool, need_cells: bool, occ: Optional[torch.Tensor] = None,
                   amax: Optional[torch.Tensor] = None) -> PillarState:
        """Band-bucketed pipeline (csrc/pillar_bands.hip): hist -> scan -> scatter -> band (4 launches; training 6)."""
        assert pts.is_cuda and pts.dtype == torch.float32 and pts.is_contiguous()
        S, N, _ = pts.shape
        dev, g, s = pts.device, self.geom, stream()
        H, W = self.H, self.W
        R, NB = self.bands(S)
        assert occ is None or (tuple(occ.shape) == (S, NB, 64) and occ.dtype == torch.int32 and occ.is_contiguous())

        def band_canvas(*a):      # the canvas-writing band call: dense, or the persistent-canvas form
            if occ is None:
                call("df_pillar2_band", *a, s)
            else:
                call("df_pillar2_band_sp", *a, ptr(occ), ptr(amax), s)
        ncol = NB + 1
        nblk = (N + call("df_pillar2_tile") - 1) // call("df_pillar2_tile")
        i32 = dict(dtype=torch.int32, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        hist = torch.empty(S, ncol, nblk, **i32)
        off = torch.empty(S, ncol, nblk, **i32)
        tot = torch.empty(S, ncol, **i32)
        counts = torch.empty(S, **i32)
        call("df_pillar2_hist", ptr(pts), S, N, g, R, ptr(hist), s)
        call("df_pillar2_scan", ptr(hist), S, ncol, nblk, ptr(off), ptr(tot), ptr(counts), s)
        points_c = torch.empty(S, N, 3, **f32)
        coords_c = torch.empty(S, N, 3, **i32)
        idx_c = torch.empty(S, N, dtype=torch.int64, device=dev)
        offs_c = torch.empty(S, N, 3, **f32)
        cpos = torch.empty(S * N, **i32)
        bkey, bidx, bpts = torch.empty(S * N, **i32), torch.empty(S * N, **i32), torch.empty(S * N, 3, **f32)
        bucket0 = torch.empty(S, NB, **i32)
        call("df_pillar2_scatter", ptr(pts), S, N, g, R, ptr(off), ptr(tot), ptr(points_c), ptr(coords_c), ptr(idx_c), ptr(offs_c),
             ptr(cpos), ptr(bkey), ptr(bidx), ptr(bpts), ptr(bucket0), s)
        key_sorted, idx_sorted, pts_sorted = torch.empty(S * N, **i32), torch.empty(S * N, **i32), torch.empty(S * N, 3, **f32)
        cell_rng = torch.empty(S * H * W, 2, **i32) if need_cells else None
        w = self._lin.weight.detach()
        SORT, STATS, CANVAS = 1, 2, 4
        if train:
            partial = torch.empty(S, NB, 32, 2, **f32)
            call("df_pillar2_band", ptr(bkey), ptr(bidx), ptr(bpts), ptr(tot), ptr(bucket0), S, g, R, SORT | STATS, ptr(w), None, 0, self.mode, out,
                 ptr(key_sorted), ptr(idx_sorted), ptr(pts_sorted), None, ptr(partial), s)
            bn_ss, bn_stride = self._bn_state(True, partial, counts, S, NB, dev)
            band_canvas(ptr(key_sorted), None, ptr(pts_sorted), ptr(tot), ptr(bucket0), S, g, R, CANVAS, ptr(w), ptr(bn_ss), bn_stride,
                        self.mode, out, None, None, None, ptr(cell_rng), None)
        else:
            bn_ss, bn_stride = self._bn_state(False, None, counts, S, NB, dev)
            band_canvas(ptr(bkey), ptr(bidx), ptr(bpts), ptr(tot), ptr(bucket0), S, g, R, SORT | CANVAS, ptr(w), ptr(bn_ss), bn_stride,
                        self.mode, out, ptr(key_sorted), ptr(idx_sorted), ptr(pts_sorted), ptr(cell_rng), None)
        return PillarState(pts, counts, points_c, coords_c, idx_c, offs_c, cpos, idx_sorted, cell_rng, key_sorted, pts_sorted, bn_ss,
                           bn_stride)

    def _sync_bn_stats(self, partial: torch.Tensor, counts: torch.Tensor) -> torch.Tensor:
        """sync_bn form of df_pfn_bn_finalize: per-sample statistics (the feature net is called once per sample) over the
        points of sample b on ALL ranks; running statistics in call order, skipped for calls with fewer than two points"""
        bn = self._bn
        sums = ops.SYNC.sum(partial.double().sum(1))                             # [B, 32, 2]
        cnt = ops.SYNC.sum(counts.double())                                      # [B]
        safe = cnt.clamp_min(1.0)[:, None]
        mean = sums[:, :, 0] / safe
        var = (sums[:, :, 1] / safe - mean * mean).clamp_min(0.0)
        invstd = torch.rsqrt(var + bn.eps)
        ga, be = bn.weight.detach().double(), bn.bias.detach().double()
        ss = torch.stack([ga * invstd, be - mean * ga * invstd, mean, invstd], 1)   # [B, 4, 32]
        ss = torch.where((cnt > 0)[:, None, None], ss, torch.zeros_like(ss)).float().contiguous()
        m = bn.momentum
        for b in range(cnt.shape[0]):
            upd = cnt[b] > 1
            unb = var[b] * (cnt[b] / (cnt[b] - 1.0).clamp_min(1.0))
            bn.running_mean.copy_(torch.where(upd, (1.0 - m) * bn.running_mean + (m * mean[b]).float(), bn.running_mean))
            bn.running_var.copy_(torch.where(upd, (1.0 - m) * bn.running_var + (m * unb).float(), bn.running_var))
        return ss

    def pillarize_bwd(self, st: PillarState, gout: DfImg, grads: Optional[Tuple[torch.Tensor, ...]]):
        """Accumulates (dW [32,9], dgamma [32], dbeta [32]) for one cloud set; grads=None starts from zero."""
        B, N, _ = st.pts.shape
        dev, g, s = st.pts.device, self.geom, stream()
        w = self._lin.weight.detach()
        nbs = max(1, min(256, (N + 31) // 32))
        acc = grads is not None
        if grads is None:
            grads = (torch.empty(32, 9, dtype=torch.float32, device=dev), torch.empty(32, dtype=torch.float32, device=dev),
                     torch.empty(32, dtype=torch.float32, device=dev))
        dW, dgamma, dbeta = grads
        partial = torch.empty(B, nbs, 32, 2, dtype=torch.float32, device=dev)
        call("df_pfn_bwd_stats", ptr(st.pts_sorted), ptr(st.cell_rng), ptr(st.key_sorted), ptr(st.counts), B, g, ptr(w), ptr(st.bn_ss),
             st.bn_stride, self.mode, gout, ptr(partial), nbs, s)
        if ops.SYNC is not None and st.bn_stride != 0:   # statistics of sample b are shared with sample b of the other ranks
            loc = partial.double().sum(1)                                         # [B, 32, (sum g, sum g * xhat)], this rank
            tb, tg = loc[:, :, 0].sum(0).float(), loc[:, :, 1].sum(0).float()
            dbeta.copy_(dbeta + tb if acc else tb)
            dgamma.copy_(dgamma + tg if acc else tg)
            cnt = ops.SYNC.sum(st.counts.double())
            glob = ops.SYNC.sum(loc.clone()) / cnt.clamp_min(1.0)[:, None, None]
            coef = torch.where((cnt > 0)[:, None, None], glob, torch.zeros_like(glob)).permute(0, 2, 1).contiguous().float()
        else:
            coef = torch.empty(B, 2, 32, dtype=torch.float32, device=dev)
            call("df_pfn_bwd_finalize", ptr(partial), B, nbs, ptr(st.counts), ptr(dgamma), ptr(dbeta), int(acc), ptr(coef), s)
            if st.bn_stride == 0:      # eval mode: running statistics are constants, no batch-statistic terms
                coef.zero_()
        dwp = torch.empty(B * nbs, 288, dtype=torch.float32, device=dev)
        call("df_pfn_bwd_weights", ptr(st.pts_sorted), ptr(st.cell_rng), ptr(st.key_sorted), ptr(st.counts), B, g, ptr(w), ptr(st.bn_ss),
             st.bn_stride, self.mode, ptr(coef), gout, ptr(dwp), nbs, s)
        call("df_colsum_finalize", ptr(dwp), B * nbs, 288, 1, ptr(dW), int(acc), s)
        return grads

    @staticmethod
    def infos_from_state(st: PillarState, counts_host: List[int]) -> List[Dict[str, torch.Tensor]]:
        return [{"points": st.points_c[b, :m], "voxel_coords": st.coords_c[b, :m], "point_idxes": st.idx_c[b, :m],
                 "point_offsets": st.offs_c[b, :m]} for b, m in enumerate(counts_host)]

    # -- reference-compatible call: embedder(points) -> (pseudoimage [B,32,H,W], infos) ---------------
    def forward(self, points: torch.Tensor):
        pts = points.contiguous().float()
        B = pts.shape[0]
        canvas = canvas_alloc(B, self.H, self.W, 32, device=pts.device)
        with torch.no_grad():
            st = self.pillarize(pts, img(canvas), self.training, need_cells=False)
        infos = self.infos_from_state(st, st.counts.tolist())
        return canvas.permute(0, 3, 1, 2), infos
