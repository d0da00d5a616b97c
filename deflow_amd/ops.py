"""Thin tensor-level wrappers over the C ABI (include/deflow_amd.h).  Every function launches HIP kernels on
torch's current stream with raw device pointers; nothing here computes with torch ops."""
from __future__ import annotations

import os

from typing import Optional, Tuple

import torch

from ._lib import DfImg, call, img, ptr, stream, ver

CONV_FWD, CONV_DGRAD = 0, 1
EPI_BIAS, EPI_STATS, EPI_BN_GELU = 0, 1, 2


def _f32(*shape, device) -> torch.Tensor:
    return torch.empty(shape, dtype=torch.float32, device=device)


def ohwi(w: torch.Tensor) -> torch.Tensor:
    """Conv weight [O,I,kh,kw] -> contiguous [O,kh,kw,I] memory (free when the parameter is channels_last)."""
    v = w.detach().permute(0, 2, 3, 1)
    if v.is_contiguous():
        v._df_keep = True       # a VIEW of the parameter: stable address, shared version counter -- the per-tensor caches may key on it
        return v
    # a parameter that is NOT channels_last (a plain nn.Module nobody converted): the contiguous copy is kept while the parameter is
    # unchanged, so that its transposes / fp16 planes are too (one copy per parameter, replaced when an optimizer writes it)
    try:
        key, epoch = (w.data_ptr(), tuple(w.shape)), (w._version, WEIGHT_GEN[0])
    except RuntimeError:
        return v.contiguous()
    hit = _OHWI_CACHE.get(key)
    if hit is not None and hit[1] == epoch:
        return hit[0]
    c = v.contiguous()
    c._df_keep = True
    if hit is not None:         # the superseded copy's transpose and planes go with it
        ck = (hit[0].data_ptr(), tuple(hit[0].shape))
        old_t = _WT_CACHE.pop(ck, None)
        if old_t is not None:
            _PLANE_CACHE.pop((old_t[0].data_ptr(), tuple(old_t[0].shape)), None)
        _PLANE_CACHE.pop(ck, None)
    if len(_OHWI_CACHE) > 256:
        _OHWI_CACHE.clear()
    _OHWI_CACHE[key] = (c, epoch, w)
    return c


_OHWI_CACHE: dict = {}


class KernelProfiler:
    """Optional per-launch HIP-event timing of the hot kernels (bench.py's live roofline figures).  Events are
    recorded on the stream the kernels are launched on (torch's current stream)."""

    def __init__(self):
        self.records = []  # (kernel / stage name, algorithmic flops, start event, end event, layer tag[, algorithmic bytes[, moved bytes]])

    def summary(self):
        out = {}
        for name, flops, e0, e1, _tag, *rest in self.records:
            d = out.setdefault(name, {"launches": 0, "flops": 0.0, "ms": 0.0, "bytes": 0.0, "moved": 0.0})
            d["launches"] += 1
            d["flops"] += flops
            d["bytes"] += rest[0] if rest else 0.0
            d["moved"] += rest[1] if len(rest) > 1 else 0.0
            d["ms"] += e0.elapsed_time(e1)
        return out


PROFILER: Optional[KernelProfiler] = None

# BatchNorm's num_batches_tracked counters: 18 one-element `add_` launches per training forward when done layer by layer.  Inside
# `deferred_tracked()` (DeFlow._run) the increments are collected and applied by ONE multi-tensor launch at the end of the forward.
_TRACKED: Optional[list] = None


def bump_tracked(t: torch.Tensor, k: int) -> None:
    if _TRACKED is None:
        t.add_(k)
    else:
        _TRACKED.append((t, int(k)))


class deferred_tracked:
    def __enter__(self):
        global _TRACKED
        self.prev = _TRACKED
        _TRACKED = []
        return self

    def __exit__(self, *exc):
        global _TRACKED
        pend, _TRACKED = _TRACKED, self.prev
        if pend:
            acc: dict = {}          # (a tensor may come twice -- the feature net's counter, once per cloud: one entry each for the launch)
            for t, k in pend:
                e = acc.setdefault(id(t), [t, 0])
                e[1] += k
            torch._foreach_add_([e[0] for e in acc.values()], [e[1] for e in acc.values()])
        return False

# Mixed-precision switch of the convolution kernels (forward, data gradient, weight gradient): True = MFMA operands rounded
# to bf16 as they leave LDS, fp32 accumulation, all tensors still fp32 (df_conv2d_mp / df_conv2d_wgrad_mp).  Set by
# optim.Trainer(dtype="bf16") around its forward + backward (BASELINE configs[4]: "bf16 MFMA" training); False everywhere else.
MFMA_BF16 = False


# bf16 STORAGE on top of the bf16-operand mode (round 3; on with Trainer(dtype="bf16") unless DF_BF16_STORE=0): inside the UNet
# encoder the conv outputs y, the activations z, and the gradients dz / dy of every 3x3 stride-1 ConvWithNorms layer live in
# HBM as bfloat16 (BatchNorm statistics from the fp32 accumulators' rounded values, normalisation / GELU / their backward in
# fp32 registers) -- the BatchNorm + GELU passes and the conv I/O move half the bytes, and the weight gradient reads bf16 tiles
# with transposing LDS reads (csrc/conv_wgrad.hip wgrad3_tr_kernel).  Stage inputs / outputs (the skip tensors) stay fp32.
BF16_STORE = False


class mfma_bf16:
    """with ops.mfma_bf16(True[, store=True]): ...  -- scoped form of the switches above"""

    def __init__(self, on: bool, store: bool = False):
        self.on, self.store = bool(on), bool(on) and bool(store)

    def __enter__(self):
        global MFMA_BF16, BF16_STORE
        self.prev, MFMA_BF16 = MFMA_BF16, self.on
        self.prev_store, BF16_STORE = BF16_STORE, self.store
        return self

    def __exit__(self, *exc):
        global MFMA_BF16, BF16_STORE
        MFMA_BF16 = self.prev
        BF16_STORE = self.prev_store
        return False


class timed:
    """with ops.timed("stage", bytes=..., flops=...): HIP events around a launch (or a short chain of launches) on the current
    stream when a KernelProfiler is installed; free otherwise.  `bytes` / `flops` are the ALGORITHMIC figures of DESIGN.md
    section 4 (what the stage has to move / compute), not what the implementation happens to do; `moved` (optional) = the bytes the
    implementation is KNOWN to stream through HBM by construction (e.g. the GRU kernels' saved planes), reported beside them."""

    def __init__(self, name: str, bytes: float = 0.0, flops: float = 0.0, tag: str = "", moved: float = 0.0):
        self.name, self.bytes, self.flops, self.tag, self.moved = name, float(bytes), float(flops), tag, float(moved)
        self.prof = PROFILER

    def __enter__(self):
        if self.prof is not None:
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.prof is not None and exc[0] is None:
            self.e1.record()
            self.prof.records.append((self.name, self.flops, self.e0, self.e1, self.tag, self.bytes, self.moved))
        return False


class SideStream:
    """Second HIP stream for the weight-gradient GEMMs of the backward pass.  They depend only on a layer's input and
    output gradient, nothing on the critical path (data gradients -> next layer) depends on them until the optimizer, so
    they run concurrently with the HBM-bound normalisation / activation backward kernels and fill the MFMA pipe while
    those stream.  `keep` pins every tensor a side-stream kernel reads until the main stream has joined (the caching
    allocator would otherwise hand the blocks to later main-stream allocations)."""

    def __init__(self, device):
        self.stream = torch.cuda.Stream(device=device)
        self.keep: list = []

    def fork(self):
        """context manager: work enqueued inside runs on the side stream, after everything enqueued so far on main"""
        ev = torch.cuda.Event()
        ev.record()
        self.stream.wait_event(ev)
        return torch.cuda.stream(self.stream)

    def join(self):
        torch.cuda.current_stream().wait_stream(self.stream)
        self.keep.clear()


SIDE: Optional[SideStream] = None
_SIDE_CACHE: dict = {}


class side_stream:
    """with ops.side_stream(on, device): ...  -- the backward inside puts its weight-gradient GEMMs on the second stream (on =
    True), keeps them on the main stream (False), or leaves the process-wide setting alone (None).  optim.Trainer turns it on
    for dtype="bf16" on one GPU: in that mode no kernel saturates the matrix pipe, so two streams overlap (52.2 -> 49.2 ms per
    step); the fp32 step, MFMA-bound end to end, loses 1 % to it and keeps one stream."""

    def __init__(self, on: Optional[bool], device=None):
        self.on, self.device = on, device

    def __enter__(self):
        global SIDE
        self.prev = SIDE
        if self.on is True:
            key = str(self.device)
            if key not in _SIDE_CACHE:
                _SIDE_CACHE[key] = SideStream(self.device)
            SIDE = _SIDE_CACHE[key]
        elif self.on is False:
            SIDE = None
        return self

    def __exit__(self, *exc):
        global SIDE
        SIDE = self.prev
        return False


def _conv_variant(x: DfImg, y: DfImg, ks: int, stride: int, mode: int, epi: int) -> str:
    """kernel template instance df_conv2d will launch (asks the library, so it cannot drift from the C dispatch)"""
    rows = y.n * y.h * y.w
    cls = mode == CONV_DGRAD and stride == 2 and ks == 3 and epi == EPI_BIAS and y.h % 2 == 0 and y.w % 2 == 0
    v = call("df_conv2d_variant", rows // 4 if cls else rows, y.grp_size * y.h * y.w, y.c, epi)
    if cls and (rows // 4) % (v // 1000):
        v = call("df_conv2d_variant", rows, y.grp_size * y.h * y.w, y.c, epi)
    bm, bn = v // 1000, v % 1000
    wm, wn = {(128, 32): (4, 1), (256, 64): (4, 1)}.get((bm, bn), (2, 2))
    kern = "conv_dma_kernel" if call("df_conv2d_last_dma") else "conv_kernel"
    halo = (kern == "conv_dma_kernel" and os.environ.get("DF_CONV_HALO", "1") != "0" and ks == 3 and stride == 1 and not cls
            and y.w % 128 == 0 and x.w == y.w and x.h == y.h and (bm, bn) in ((128, 128), (128, 64)))
    if halo:  # haloed-A kernel (mirrors df_conv2d's dispatch)
        return f"conv_halo_kernel<{bn},{2 if bn == 128 else 4},{4 if bn == 128 else 2}>"
    if kern == "conv_dma_kernel":  # 8-wave forms of the two big tiles (DF_CONV_W8 bit 0 / bit 1, default both)
        w8 = int(os.environ.get("DF_CONV_W8", "3"))
        if (bm, bn) == (128, 128) and (w8 & 1):
            wm, wn = 2, 4
        elif (bm, bn) == (128, 64) and (w8 & 2):
            wm, wn = 4, 2
    return f"{kern}<{bm},{bn},{wm},{wn}>"


# ---- max |x| bookkeeping of the fp16x2 convolution kernels (df_conv2d_h2 / df_conv2d_wgrad_h2) -------------------------------
# The kernels scale each operand by a power of two taken from an UPPER BOUND of its absolute maximum (a device scalar, so nothing
# here synchronises); any bound within a factor of ~2^10 serves.  Where the bound comes from:
#   activations / gradients  written by the BatchNorm + GELU passes: measured by those kernels as they write (bn_gelu_apply,
#                            bn_gelu_bwd); everything else: one df_absmax pass the first time a convolution reads the tensor,
#                            remembered on the descriptor / tensor object (the tape keeps it for the weight gradient)
#   weights                  ops.W_AMAX when a Trainer set it (one df_absmax over the whole parameter arena per step), else
#                            per call
# Slots come zero-filled from a small pool (one fill per _AMAX_POOL_N slots; amax_pool_reset() starts a fresh pool -- the Trainer
# does so at the top of every step, inside a captured step too, so that a replay re-zeroes what it accumulates into).
_AMAX_POOL_N = 256
_amax_pool = [None, 0]
W_AMAX: Optional[torch.Tensor] = None


def amax_pool_reset():
    _amax_pool[0] = None


def amax_slot(device) -> torch.Tensor:
    if _amax_pool[0] is None or _amax_pool[1] >= _AMAX_POOL_N or _amax_pool[0].device != torch.device(device):
        _amax_pool[0] = torch.zeros(_AMAX_POOL_N, dtype=torch.float32, device=device)
        _amax_pool[1] = 0
    i = _amax_pool[1]
    _amax_pool[1] = i + 1
    return _amax_pool[0][i:i + 1]


def amax_of(d: DfImg, device) -> torch.Tensor:
    """device scalar >= max |x| of the image view: the producer's measurement if it left one, else one df_absmax pass"""
    a = getattr(d, "_amax", None)
    if a is None:
        a = amax_slot(device)
        call("df_absmax", d, ptr(a), stream())
        d._amax = a
        src = getattr(d, "_src", None)
        if src is not None:               # whole-tensor view: the next descriptor made of the tensor finds the measurement (_lib.img)
            src._df_amax = (a, ver(src))
    return a


def wrote(t: torch.Tensor) -> None:
    """a kernel wrote (accumulated into) t through a raw pointer, without measuring: whatever bound t carried is no longer one
    (the kernels leave torch's version counter alone, so the record would otherwise stay 'valid' -- ADVICE r3)"""
    if getattr(t, "_df_amax", None) is not None:
        t._df_amax = None


def _h2_on() -> bool:
    return os.environ.get("DF_CONV_H2", "1") != "0"


# ---- PRE-SPLIT ("h2") tensors, round 4 (include/deflow_amd.h "PRE-SPLIT tensors"; csrc/elementwise.hip) -------------------------
# An activation / gradient whose only consumers are the fp16x2 3x3 convolution and weight-gradient kernels is stored by its PRODUCER as
# two scaled fp16 planes (the bytes of the fp32 tensor), with the scale taken from a bound of max |x| that is known before the producer
# writes: BatchNorm + GELU outputs from the statistics (df_bn_finalize2 / df_bn_bwd_finalize2), plain convolution outputs from
# max |input| x the largest L1 norm of a weight row (+ max |bias|).  The consumers then fetch their operands by LDS-DMA with no
# in-kernel split (conv_halo_x3_kernel<.., XP>, wgrad3_h2p_kernel).  DF_H2P=0 keeps every tensor fp32 (round 3's data flow).
def h2p_on() -> bool:
    return h2_active() and os.environ.get("DF_H2P", "1") != "0"


def h2_empty(shape, device, bound: torch.Tensor) -> torch.Tensor:
    """storage of an h2 image [N,H,W,C] (C % 32 == 0): float32-typed memory holding per pixel and 32-channel chunk [32 fp16 hi | 32
    fp16 lo]; `bound` = the device scalar that defines its power-of-two scale (every producer and consumer takes THIS tensor)."""
    assert shape[-1] % 32 == 0
    t = torch.empty(shape, dtype=torch.float32, device=device)
    t._df_h2 = bound
    return t


def h2_unpack(t: torch.Tensor) -> torch.Tensor:
    """fp32 copy of an h2 tensor (tests / debugging)"""
    out = torch.empty(t.shape, dtype=torch.float32, device=t.device)
    call("df_h2_unpack", img(t), ptr(t._df_h2), img(out), stream())
    return out


def h2_pack(x: torch.Tensor, bound: torch.Tensor) -> torch.Tensor:
    t = h2_empty(x.shape, x.device, bound)
    call("df_h2_pack", img(x), ptr(bound), img(t), stream())
    return t


class WeightPrep:
    """Every convolution layer's per-step weight forms from ONE launch (df_weight_prep; round 4): transposed weights, [hi | lo] fp16
    planes of the weights and of their transpose (every layer: the 3x3 stride-1 kernels and, since the round's second session, the
    1x1 / stride-2 kernel read them), row L1 norms and max |bias| (a-priori bounds of pre-split outputs).  optim.Trainer builds one at the top of every step (the parameters change once per step) and installs it as
    ops.WPREP; ops.weight_transpose / _split_h2 / rows_l1max answer from it when the tensor they are asked about is one of its
    layers, and fall back to their own launches otherwise (plain autograd users, the decoder head's GEMM weights)."""

    def __init__(self, convs, w_amax: torch.Tensor):
        import numpy as np
        self.convs = [m for m in convs if m.weight.dim() == 4]
        dev = self.convs[0].weight.device
        ws = [ohwi(m.weight) for m in self.convs]
        assert all(w.data_ptr() == m.weight.data_ptr() for w, m in zip(ws, self.convs)), "conv weights must be channels_last (OHWI memory)"
        ptrs = [w.data_ptr() for w in ws] + [m.bias.data_ptr() for m in self.convs if m.bias is not None]
        self.base = min(ptrs)
        rec = np.zeros(len(ws), dtype=[("w_off", "<i8"), ("b_off", "<i8"), ("wt_off", "<i8"), ("w2_off", "<i8"), ("wt2_off", "<i8"),
                                       ("cout", "<i4"), ("taps", "<i4"), ("cin", "<i4"), ("split", "<i4"), ("blk0", "<i4"), ("pad", "<i4")])
        off, blk = 0, 0
        self.layer = {}
        for i, (w, m) in enumerate(zip(ws, self.convs)):
            co, kh, kw, ci = w.shape
            n = w.numel()
            split = 1      # (round 4, second session: the 1x1 / stride-2 layers read pre-split weights too: conv_dma_kernel<.., H2, BP>)
            rec[i] = ((w.data_ptr() - self.base) // 4, (m.bias.data_ptr() - self.base) // 4 if m.bias is not None else -1,
                      off, off + n, off + 2 * n, co, kh * kw, ci, split, blk, 0)
            self.layer[w.data_ptr()] = (i, off, n, (co, kh, kw, ci), split)
            off += 3 * n if split else n
            blk += co + ci
        self.total_blocks, self.nl = blk, len(ws)
        self.table = torch.from_numpy(rec.view(np.uint8).copy()).to(dev)
        self.out = torch.empty(off, dtype=torch.float32, device=dev)
        self.norms = torch.zeros(self.nl, 3, dtype=torch.float32, device=dev)
        self.w_amax = w_amax
        self.by_wt = {}                 # data_ptr of a transposed view -> (layer, plane offset) for _split_h2 of the transpose
        self.bias_of = {m.bias.data_ptr(): self.layer[w.data_ptr()][0] for w, m in zip(ws, self.convs) if m.bias is not None}
        self._anchor = torch.empty(0, dtype=torch.float32, device=dev)

    def stale(self) -> bool:
        """True when a conv parameter no longer lives where the table says (model.to(), load_state_dict(assign=True), p.data = ...):
        the table holds raw addresses, so the caller must rebuild the WeightPrep before running it"""
        for m in self.convs:
            e = self.layer.get(m.weight.data_ptr())          # (the OHWI view of a channels_last weight starts at the same address)
            co, ci, kh, kw = m.weight.shape
            if e is None or e[3] != (co, kh, kw, ci):
                return True
            if m.bias is not None and m.bias.data_ptr() not in self.bias_of:
                return True
        return False

    def run(self, w_amax: torch.Tensor):
        """(re)compute every form for the current parameter values"""
        import ctypes
        assert not self.stale(), "WeightPrep: a conv parameter moved since the table was built -- rebuild it (Trainer does: _wprep)"
        self.w_amax = w_amax
        self.norms.zero_()
        call("df_weight_prep", ctypes.c_void_p(self.base), ptr(self.table), self.nl, self.total_blocks, ptr(w_amax), ptr(self.out), ptr(self.norms),
             stream())

    def wt(self, w_ohwi: torch.Tensor) -> Optional[torch.Tensor]:
        e = self.layer.get(w_ohwi.data_ptr())
        if e is None:
            return None
        i, off, n, (co, kh, kw, ci), split = e
        t = self.out[off:off + n].view(ci, kh, kw, co)
        self.by_wt[t.data_ptr()] = (i, off + 2 * n, n, split)
        return t

    def h2(self, w: torch.Tensor):
        """[hi | lo] planes (a float16 view of 2 x numel) of a layer's weights, or of its transpose as self.wt() returned it"""
        e = self.layer.get(w.data_ptr())
        if e is not None and e[4]:
            _, off, n, _, _ = e
            return self.out[off + n:off + 2 * n].view(torch.float16), self.w_amax
        e = self.by_wt.get(w.data_ptr())
        if e is not None and e[3]:
            _, o2, n, _ = e
            return self.out[o2:o2 + n].view(torch.float16), self.w_amax
        return None

    def l1(self, w: torch.Tensor, bias: Optional[torch.Tensor]):
        """(max row L1 norm, max |bias| or None) device scalars for a layer's weights or their transpose; None if not one of ours"""
        e = self.layer.get(w.data_ptr())
        col = 0
        if e is None:
            e2 = self.by_wt.get(w.data_ptr())
            if e2 is None:
                return None
            i, col = e2[0], 1
        else:
            i = e[0]
        bm = None
        if bias is not None:
            j = self.bias_of.get(bias.data_ptr())
            if j is None:
                return None
            bm = self.norms[j, 2:3]
        return self.norms[i, col:col + 1], bm


WPREP: Optional[WeightPrep] = None


def rows_l1max(w2d_rows: int, row_len: int, w: torch.Tensor, bias: Optional[torch.Tensor] = None):
    """-> (max_row sum |w[row, :]|, max |bias| or None) as device scalars: the weight side of a conv output's a-priori bound"""
    if WPREP is not None:
        hit = WPREP.l1(w, bias)
        if hit is not None:
            return hit
    l1 = amax_slot(w.device)
    bm = amax_slot(w.device) if bias is not None else None
    call("df_rows_l1max", ptr(w), w2d_rows, row_len, ptr(bias), 0 if bias is None else bias.numel(), ptr(l1), ptr(bm), stream())
    return l1, bm


def h2_bound(a: torch.Tensor, l1: Optional[torch.Tensor] = None, b: Optional[torch.Tensor] = None,
             other: Optional[torch.Tensor] = None, slack: float = 1.001) -> torch.Tensor:
    """device scalar max(other, a * l1 * slack + b)"""
    out = torch.empty(1, dtype=torch.float32, device=a.device)
    call("df_h2_bound", ptr(out), ptr(a), ptr(l1), ptr(b), ptr(other), float(slack), stream())
    return out


def conv_out_bound(x: DfImg, w_rows: torch.Tensor, bias: Optional[torch.Tensor], device, other: Optional[torch.Tensor] = None) -> torch.Tensor:
    """bound of max |conv(x; w) + bias| for the kernel operand `w_rows` ([rows = output channels][taps x input channels] memory:
    the OHWI weights of a forward conv, the transposed ones of a data gradient):  max|x| x max_row ||w_row||_1 + max |bias|"""
    rows = w_rows.shape[0]
    l1, bm = rows_l1max(rows, w_rows.numel() // rows, w_rows, bias)
    return h2_bound(amax_of(x, device), l1, bm, other)


def h2_active() -> bool:
    """are the fp32-mode 3x3 stride-1 convolutions on the fp16x2 kernels (so that producers should measure max |x|)?"""
    return (not MFMA_BF16) and _h2_on() and os.environ.get("DF_CONV_X3", "1") != "0"


_UP_FUSED_NO: set = set()     # (shape keys for which df_conv2d_h2f_wp_up answered DF_E_SHAPE: the two launches are used)


def conv1x1_up_fused(x: DfImg, w_ohwi: torch.Tensor, bias: torch.Tensor, y: DfImg, t: DfImg, align_corners: bool) -> bool:
    """The 1x1 skip convolution of an UpsampleSkip block into the SECOND half `y` of a pre-split concatenation, its workgroups also
    writing the first half = bilinear x2 of `t` (df_conv2d_h2f_wp_up, round 6): -> True if that one launch was issued, False if the
    call is not covered (no WeightPrep planes / no bound on x / shape outside the 8-wave DMA kernels) -- the caller then issues
    ops.upsample2x + ops.conv2d.  DF_UP_FUSED=0: never."""
    if (os.environ.get("DF_UP_FUSED", "1") == "0" or not h2_active() or MFMA_BF16 or y.elt != 2 or x.elt != 0 or t.elt != 0
            or getattr(x, "_amax", None) is None or getattr(y, "_amax", None) is None):
        return False
    wp = _wprep_planes(w_ohwi)
    if wp is None:
        return False
    key = (x.n, x.h, x.w, x.c, y.c, y.ld)
    if key in _UP_FUSED_NO:
        return False
    prof = PROFILER
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    try:
        call("df_conv2d_h2f_wp_up", x, ptr(w_ohwi), ptr(wp[0]), ptr(x._amax), ptr(wp[1]), ptr(bias), y, ptr(y._amax), t, int(align_corners), stream())
    except RuntimeError as e:
        if "DF_E_SHAPE" not in str(e):
            raise
        _UP_FUSED_NO.add(key)
        return False
    if prof is not None:
        e1.record()
        prof.records.append(("conv_dma_kernel<up>/h2", 2.0 * y.n * y.h * y.w * x.c * y.c, e0, e1,
                             f"fwd 1x1 s1 {x.c}->{y.c} @{y.h}x{y.w} x{y.n} fh + bilinear x2", float(x.n * x.h * x.w * x.c * 4 + 2 * y.n * y.h * y.w * y.c * 4)))
    return True


def _split_h2(w_ohwi: torch.Tensor):
    """weights -> (two fp16 planes, their amax): per call (they change every optimizer step; each conv uses them once per direction)
    -- or from the step's WeightPrep when the tensor is one of its layers (or a transpose it handed out)"""
    if WPREP is not None:
        hit = WPREP.h2(w_ohwi)
        if hit is not None:
            return hit
    wa = W_AMAX
    if wa is None:
        # outside a trainer step (inference, plain autograd): planes are kept per weight tensor until it changes -- an eval-mode forward
        # split every 3x3 layer's weights on every call (round 5: 14 df_split_h2 + 21 df_absmax launches of the 95 of a B = 1 forward)
        # ONE entry per weight tensor (keyed by address and shape, validated by its write epoch): a changed tensor replaces its entry,
        # so plain-autograd training does not pile up a triple per layer and step until a clear
        key, epoch = _cache_key(w_ohwi)
        hit = _PLANE_CACHE.get(key) if key is not None else None
        if hit is not None and hit[3] == epoch:
            return hit[0], hit[1]
        # (a bound of its own tensor, NOT a pooled slot: amax_pool_reset() recycles those between steps)
        wa = torch.zeros(1, dtype=torch.float32, device=w_ohwi.device)
        call("df_absmax", img(w_ohwi.reshape(1, 1, -1, w_ohwi.shape[-1])), ptr(wa), stream())
        w2 = torch.empty(2 * w_ohwi.numel(), dtype=torch.float16, device=w_ohwi.device)
        call("df_split_h2", ptr(w_ohwi), ptr(wa), ptr(w2), w_ohwi.numel(), stream())
        if key is not None:
            if len(_PLANE_CACHE) > 256:
                _PLANE_CACHE.clear()
            _PLANE_CACHE[key] = (w2, wa, w_ohwi, epoch)     # (the tensor is kept: its address cannot be reused while cached)
        return w2, wa
    w2 = torch.empty(2 * w_ohwi.numel(), dtype=torch.float16, device=w_ohwi.device)
    call("df_split_h2", ptr(w_ohwi), ptr(wa), ptr(w2), w_ohwi.numel(), stream())
    return w2, wa


_PLANE_CACHE: dict = {}
_WT_CACHE: dict = {}


def _cache_key(t: torch.Tensor):
    """-> ((address, shape), write epoch) of a weight tensor for the per-tensor caches of non-trainer callers; (None, None) for
    tensors that do not track versions (inference mode)"""
    # only tensors MARKED as having a stable identity (`_df_keep`): ops.ohwi's view of a channels_last parameter and the transposes
    # weight_transpose keeps for those.  Anything else -- a per-call contiguous copy, the decoder's packed gate matrices, a gradient
    # passed through weight_transpose -- is a fresh tensor whose address may repeat with version 0: never cached
    if not getattr(t, "_df_keep", False):
        return None, None
    try:
        return (t.data_ptr(), tuple(t.shape)), (t._version, WEIGHT_GEN[0])
    except RuntimeError:
        return None, None


def _wprep_planes(w_ohwi: torch.Tensor):
    """(planes, their amax) of a 1x1 / stride-2 layer's weights (or of the transpose ops.weight_transpose handed out) if the step's
    WeightPrep made them, else from a split of its own (kept while the tensor is unchanged) -- conv_dma_kernel<.., H2, BP> fetches the
    weights pre-split.  DF_CONV_H2F_WP=0: None (the kernel splits its weight fragments itself)."""
    if os.environ.get("DF_CONV_H2F_WP", "1") == "0":
        return None
    if WPREP is not None:
        hit = WPREP.h2(w_ohwi)
        if hit is not None:
            return hit
    if W_AMAX is not None:      # inside a trainer step without the weight prep (DF_WPREP=0): the same kernel, planes from a per-call split
        return _split_h2(w_ohwi)
    # plain autograd / inference callers: the same kernel again (one form of the convolution whoever calls it: the captured trainer
    # program and a hand-driven eager step agree bit for bit, tests/helpers/rccl_world1.py); planes are kept per weight tensor until it
    # changes (inference: split once; a transposed copy made per call misses and is split per call)
    return _split_h2(w_ohwi)     # (W_AMAX is None here: _split_h2 keeps the planes in _PLANE_CACHE)


def conv2d(x: DfImg, w_ohwi: torch.Tensor, bias: Optional[torch.Tensor], y: DfImg, ks: int, stride: int = 1,
           mode: int = CONV_FWD, epi: int = EPI_BIAS, scale=None, shift=None, stats=None, accumulate: bool = False,
           amax_out: Optional[torch.Tensor] = None, bwd_bn=None):
    """amax_out: the slot the output's max |y| is accumulated into (fp16x2 mode; default: a fresh one) -- several producers of one
    buffer (the two halves of a concatenation) share a slot.
    bwd_bn = (y_prev, bn_ss_prev, partial): a 3x3 stride-1 DATA gradient whose output is the gradient of a BatchNorm + GELU layer's
    output -- the fp16x2 kernel's epilogue then also leaves that layer's backward partial sums in `partial` (df_conv2d_h2p_dgrad_bn).
    -> True if it did (the caller then skips ops.bn_gelu_bwd's reduce pass); ignored (False) on any other kernel."""
    prof = PROFILER
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    w16 = (MFMA_BF16 and ks == 3 and stride == 1 and os.environ.get("DF_CONV_W16", "1") != "0"
           and call("df_conv2d_w16_ok", x, y, ks, stride, mode, epi) == 1)
    pre = x.elt == 2 or y.elt == 2     # pre-split input and / or output (h2 images; their bounds ride on the descriptors)
    if pre:
        assert not MFMA_BF16 and _h2_on() and not accumulate
        if ks == 3 and stride == 1:
            assert call("df_conv2d_h2p_ok", x, y, ks, stride, mode, epi) == 1, "no pre-split tile form for this shape (ask df_conv2d_h2p_ok first)"
        else:
            assert x.elt == 0, "1x1 / stride-2 convolutions read fp32"
    x3 = (not MFMA_BF16 and not w16 and ks == 3 and stride == 1 and ((x.elt == 0 and y.elt == 0) or pre)
          and (pre or call("df_conv2d_x3_ok", x, y, ks, stride, mode, epi) == 1))
    h2 = x3 and _h2_on()
    # fp32 mode with the fp16x2 kernels on: every fp32 conv output carries its max |y| for a possible fp16x2 consumer (measured by
    # the epilogue; outputs that go through BatchNorm + GELU first get theirs from that pass instead)
    ya = None
    if h2_active() and y.elt == 0 and epi == EPI_STATS and amax_out is not None and not w16:
        ya = amax_out          # max |y| of a conv output that BatchNorm + GELU follow: the bounds of the pre-split z / dy come from it
    elif h2_active() and y.elt == 0 and epi != EPI_STATS and not w16:
        ya = amax_out if amax_out is not None else amax_slot(w_ohwi.device)
        y._amax = ya
        src = getattr(y, "_src", None)      # the descriptor covers a whole tensor: descriptors made of it later inherit the bound
        if src is not None:
            src._df_amax = (ya, ver(src))
    if ya is None and getattr(y, "_src", None) is not None:
        wrote(y._src)          # (an unmeasured write into a tensor that carried a bound)
    # (round 5: also the 3x3 stride-1 layers that have NO haloed fp16x2 tile form -- too few pixels: the 64 x 64 layers of a B = 1
    # forward ran on the fp32 MFMA, 5 x 94 us of a 2.2 ms forward)
    h2f = (h2_active() and not x3 and not w16 and x.elt == 0 and y.elt in (0, 2) and (ks == 1 or stride == 2 or (ks == 3 and stride == 1))
           and x.c % (64 if os.environ.get("DF_CONV_H2F_C32", "1") == "0" else 32) == 0     # (32: the first encoder conv on the canvas: its bound comes from the pillar feature net's statistics)
           and getattr(x, "_amax", None) is not None and os.environ.get("DF_CONV_H2F", "1") != "0")
    fused_bn = False
    if (h2 and bwd_bn is not None and mode == CONV_DGRAD and y.elt == 0 and not accumulate and bias is None and y.ld == y.c
            and os.environ.get("DF_FUSE_BN_BWD", "1") != "0" and call("df_conv2d_h2p_ok", x, y, ks, stride, mode, 3) == 1):
        w2, wa = _split_h2(w_ohwi)
        yp, ssp, part = bwd_bn
        call("df_conv2d_h2p_dgrad_bn", x, ptr(w2), ptr(amax_of(x, w_ohwi.device)), ptr(wa), y, ptr(yp), ptr(ssp), ptr(part), ptr(ya), stream())
        fused_bn = True
    elif h2 and pre:
        w2, wa = _split_h2(w_ohwi)
        call("df_conv2d_h2p", x, ptr(w2), ptr(amax_of(x, w_ohwi.device)), ptr(wa), ptr(bias), y, ptr(y._amax) if y.elt == 2 else None,
             ks, stride, ks // 2, mode, epi, ptr(scale), ptr(shift), ptr(stats), 0, ptr(ya), stream())
    elif pre:      # 1x1 (or stride-2) convolution with a pre-split OUTPUT: the fp32-input kernels, fp16x2 on the fragments when x has a bound
        wa = None
        wp = _wprep_planes(w_ohwi) if h2f else None
        if h2f and wp is None:      # (the in-kernel weight split, DF_CONV_H2F_WP=0: needs the weights' bound)
            wa = W_AMAX
            if wa is None:
                wa = amax_slot(w_ohwi.device)
                call("df_absmax", img(w_ohwi.reshape(1, 1, -1, w_ohwi.shape[-1])), ptr(wa), stream())
        if wp is not None:      # the step's WeightPrep holds this layer's planes: no weight split in the kernel (conv_dma_kernel<.., H2, BP>)
            call("df_conv2d_h2f_wp", x, ptr(w_ohwi), ptr(wp[0]), ptr(x._amax), ptr(wp[1]), ptr(bias), y, ptr(y._amax), ks, stride, ks // 2, mode,
                 epi, ptr(scale), ptr(shift), ptr(stats), 0, None, stream())
        else:
            call("df_conv2d_yh2", x, ptr(w_ohwi), ptr(x._amax) if h2f else None, ptr(wa), ptr(bias), y, ptr(y._amax), ks, stride, ks // 2, mode, epi,
                 ptr(scale), ptr(shift), ptr(stats), 0, stream())
    elif h2:
        # fp32-accurate product from TWO fp16 planes per operand with per-tensor power-of-two scales (conv_halo_x3_kernel<NP = 2>:
        # three MFMAs per operand pair instead of six)
        w2, wa = _split_h2(w_ohwi)
        call("df_conv2d_h2", x, ptr(w2), ptr(amax_of(x, w_ohwi.device)), ptr(wa), ptr(bias), y, ks, stride, ks // 2, mode, epi,
             ptr(scale), ptr(shift), ptr(stats), int(accumulate), ptr(ya), stream())
    elif h2f:
        # 1x1 / stride-2 convolutions whose input already carries a bound: fp16x2 on the fragments of the DMA-tile kernel
        wp = _wprep_planes(w_ohwi)
        wa = W_AMAX
        if wp is None and wa is None:      # (the in-kernel weight split, DF_CONV_H2F_WP=0: needs the weights' bound)
            wa = amax_slot(w_ohwi.device)
            call("df_absmax", img(w_ohwi.reshape(1, 1, -1, w_ohwi.shape[-1])), ptr(wa), stream())
        if wp is not None:
            call("df_conv2d_h2f_wp", x, ptr(w_ohwi), ptr(wp[0]), ptr(x._amax), ptr(wp[1]), ptr(bias), y, None, ks, stride, ks // 2, mode, epi,
                 ptr(scale), ptr(shift), ptr(stats), int(accumulate), ptr(ya), stream())
        else:
            call("df_conv2d_h2f", x, ptr(w_ohwi), ptr(x._amax), ptr(wa), ptr(bias), y, ks, stride, ks // 2, mode, epi, ptr(scale), ptr(shift),
                 ptr(stats), int(accumulate), ptr(ya), stream())
    elif ya is not None and not x3:
        call("df_conv2d_amax", x, ptr(w_ohwi), ptr(bias), y, ks, stride, ks // 2, mode, epi, ptr(scale), ptr(shift), ptr(stats),
             int(accumulate), ptr(ya), stream())
    elif x3:
        # fp32-accurate product from three bf16 planes per operand (conv_halo_x3_kernel): the weights are split once per call
        # (they change every optimizer step), the activations in the kernel's staging
        w3 = torch.empty(3 * w_ohwi.numel(), dtype=torch.bfloat16, device=w_ohwi.device)
        call("df_split_bf16x3", ptr(w_ohwi), ptr(w3), w_ohwi.numel(), stream())
        call("df_conv2d_x3", x, ptr(w3), ptr(bias), y, ks, stride, ks // 2, mode, epi, ptr(scale), ptr(shift), ptr(stats),
             int(accumulate), stream())
    elif w16:
        # bf16 tiles in LDS (conv_halo_w16_kernel): the weights are cast once per call (they change every optimizer step and
        # each conv uses them once per direction), the activations stay fp32 in memory
        wb = torch.empty(w_ohwi.numel(), dtype=torch.bfloat16, device=w_ohwi.device)
        call("df_cast_bf16", ptr(w_ohwi), ptr(wb), w_ohwi.numel() // x.c, x.c, x.c, x.c, stream())
        call("df_conv2d_w16", x, ptr(wb), ptr(bias), y, ks, stride, ks // 2, mode, epi, ptr(scale), ptr(shift), ptr(stats),
             int(accumulate), stream())
    else:
        call("df_conv2d_mp", x, ptr(w_ohwi), ptr(bias), y, ks, stride, ks // 2, mode, epi, ptr(scale), ptr(shift), ptr(stats),
             int(accumulate), int(MFMA_BF16), stream())
    if prof is not None:
        e1.record()
        small = y if mode == CONV_FWD else x  # the conv-output-sized grid
        flops = 2.0 * small.n * small.h * small.w * ks * ks * x.c * y.c
        tag = f"{'fwd' if mode == CONV_FWD else 'dgrad'} {ks}x{ks} s{stride} {x.c}->{y.c} @{small.h}x{small.w} x{small.n} {'fbh'[x.elt]}{'fbh'[y.elt]}"
        name = _conv_variant(x, y, ks, stride, mode, epi) + ("/h2" if (h2f and "conv_dma_kernel<128," in _conv_variant(x, y, ks, stride, mode, epi)) else "")
        if x3:   # mirrors conv2d_impl's dispatch of the bf16x3 forms (BM, BN, WM, WN, SEG, DB)
            bn = 128 if y.c % 128 == 0 else 64
            seg = 1 if y.w % 128 == 0 else 2
            m_rows, rpg = y.n * y.h * y.w, y.grp_size * y.h * y.w
            bm = 256 if (bn == 64 and seg == 1 and y.w % 256 == 0 and m_rows % 256 == 0 and (epi != EPI_STATS or rpg % 256 == 0)
                         and os.environ.get("DF_CONV_X3_BM256", "1") != "0") else 128
            name = f"conv_halo_x3_kernel<{bm},{bn},{2 if bn == 128 else 4},{4 if bn == 128 else 2},{seg},{4 if (bn == 128 or bm == 256) else 8}{',2' if h2 else ''}>"
            seg2 = 1 if y.w % 256 == 0 else 2 if y.w == 128 else 4 if y.w == 64 else 0
            if (h2 and bn == 128 and seg2 and y.h % seg2 == 0 and m_rows % 256 == 0 and (epi != EPI_STATS or rpg % 256 == 0)
                    and m_rows // 256 * (y.c // 128) >= 512 and os.environ.get("DF_CONV_H2_BM256", "1") != "0"):
                name = f"conv_halo_x3_kernel<256,128,4,2,{seg2},4,2>"     # 64 x 64 wave tiles
            seg64 = 1 if y.w % 512 == 0 else 2 if y.w == 256 else 0
            if (h2 and bn == 64 and seg64 and y.h % seg64 == 0 and m_rows % 512 == 0 and (epi != EPI_STATS or rpg % 512 == 0)
                    and m_rows // 512 >= 512 and os.environ.get("DF_CONV_H2_BM256", "1") != "0"):
                name = f"conv_halo_x3_kernel<512,64,8,1,{seg64},3,2>"
            if x.elt == 2:
                name = name[:-1] + ",xp>"        # pre-split input: halo by LDS-DMA (template argument XP)
        if w16:
            bn = 128 if y.c % 128 == 0 else 64
            name = f"conv_halo_w16_kernel<{bn},{2 if bn == 128 else 4},{4 if bn == 128 else 2},{1 if y.w % 128 == 0 else 2}>"
            m_rows, rpg = y.n * y.h * y.w, y.grp_size * y.h * y.w
            wide = os.environ.get("DF_W16_WIDE", "0") == "1" and m_rows > 8192
            seg2 = 1 if y.w % 256 == 0 else 2 if y.w == 128 else 4 if y.w == 64 else 0
            seg64 = 1 if y.w % 512 == 0 else 2 if y.w == 256 else 0
            if (wide and bn == 128 and seg2 and y.h % seg2 == 0 and m_rows % 256 == 0 and (epi != EPI_STATS or rpg % 256 == 0)
                    and m_rows // 256 * (y.c // 128) >= 512):
                name = f"conv_halo_x3_kernel<256,128,4,2,{seg2},8,1,{'true' if x.elt else 'false'}>"     # one bf16 plane, 64 x 64 wave tiles
            elif (wide and bn == 64 and seg64 and y.h % seg64 == 0 and m_rows % 512 == 0 and (epi != EPI_STATS or rpg % 512 == 0)
                    and m_rows // 512 >= 512):
                name = f"conv_halo_x3_kernel<512,64,8,1,{seg64},8,1,{'true' if x.elt else 'false'}>"
        # operand bytes of the call (input + output images once, the weights once): what an HBM-bound small-K layer is priced against
        esz = lambda d: (4, 2, 4)[d.elt]
        obytes = float(x.n * x.h * x.w * x.c * esz(x) + y.n * y.h * y.w * y.c * esz(y) * (2 if accumulate else 1) + w_ohwi.numel() * 4)
        prof.records.append((name + ("/bf16" if MFMA_BF16 else ""), flops, e0, e1, tag, obytes))
    return fused_bn


def conv_tile_m(rows_per_group: int, cout: int) -> int:
    return call("df_conv2d_tile_m", rows_per_group, cout)


# Folded eval-mode BatchNorm (scale, shift, mean, invstd) is cached per module: five tiny torch launches per layer and
# forward otherwise -- 90 of the 175 launches of a B=1 inference.  The cache key carries the tensors' version counters
# plus PARAM_GEN, which every HIP-side writer that bypasses autograd's counters bumps (Adam on the flat arena, the
# running-statistics update of a training forward).
PARAM_GEN = [0]
# ... and WEIGHT_GEN, bumped only where WEIGHTS are written that way (the arena's Adam kernel, a captured step's replay): the epoch of the
# per-tensor weight caches (ohwi copies, transposes, fp16 planes) -- the running-statistics updates of a training forward bump
# PARAM_GEN eighteen times per step and would void them for nothing
WEIGHT_GEN = [0]


def invalidate_weight_caches() -> None:
    """Void every per-tensor weight cache (fp16 planes, transposes, OHWI copies) and every folded BatchNorm.  The caches are validated by
    (tensor._version, WEIGHT_GEN): writers that go through autograd's version counters -- optimizer.step() in eager PyTorch,
    load_state_dict, parameter.copy_() -- need nothing.  Writers that BYPASS the counters must call this (or bump the generations) after
    they write: a user-captured CUDA-graph replay of a torch optimizer step, raw-pointer writes into the parameter arena, another
    library updating the weights in place.  FlatAdam.step / step_captured and the trainer's graph replay do it themselves;
    load_from_checkpoint calls it for good measure."""
    WEIGHT_GEN[0] += 1
    PARAM_GEN[0] += 1


def folded_bn(bn) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    key = (PARAM_GEN[0], bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version,
           bn.weight.data_ptr(), bn.running_var.data_ptr(), bn.eps)
    c = getattr(bn, "_df_fold", None)
    if c is None or c[0] != key:
        invstd = torch.rsqrt(bn.running_var + bn.eps)
        scale = (bn.weight.detach() * invstd).contiguous()
        shift = (bn.bias.detach() - bn.running_mean * scale).contiguous()
        c = (key, scale, shift, bn.running_mean, invstd)
        bn._df_fold = c
    return c[1], c[2], c[3], c[4]


class SyncBN:
    """sync_bn (Lightning's sync_batchnorm / torch SyncBatchNorm) for the data-parallel ranks: the per-channel sums behind a
    BatchNorm's batch statistics -- forward (sum, sum of squares, count) and backward (sum of g, sum of g * xhat) -- are
    all-reduced between the partial-sum kernels and the finalisation, which then runs as a few fp64 torch ops on [C]-sized
    tensors instead of the fused finalize kernels.  Parameter gradients (dgamma, dbeta) stay rank-local sums: the
    gradient all-reduce averages them like every other parameter (as torch's SyncBatchNorm + DDP do)."""

    def __init__(self, dist, group, world: int):
        self.dist, self.group, self.world = dist, group, world

    def sum(self, t: torch.Tensor) -> torch.Tensor:
        self.dist.all_reduce(t, group=self.group)
        return t


SYNC: Optional[SyncBN] = None   # set by optim.Trainer(sync_bn=True) on a multi-rank process group


def bn_finalize(partial, tiles_per_group, groups, C, count, gamma, beta, eps, momentum, rmean, rvar, bn_ss, y_amax=None, z_bound=None):
    """y_amax / z_bound (both or neither): max |y| as the conv epilogue measured it, and the zero-initialised slot that receives the
    bound of max |gelu(bn(y))| -- the scale of the pre-split z the apply pass writes next"""
    PARAM_GEN[0] += 1  # running statistics change under any cached eval-mode fold
    if SYNC is not None:
        red = SYNC.sum(partial.view(groups, tiles_per_group, C, 2).to(torch.float64).sum(1))      # [groups, C, 2], all ranks
        cnt = float(count) * SYNC.world
        ga = gamma.double() if gamma is not None else torch.ones(C, dtype=torch.float64, device=partial.device)
        be = beta.double() if beta is not None else torch.zeros(C, dtype=torch.float64, device=partial.device)
        out = bn_ss.view(groups, 4, C)
        for g in range(groups):                    # groups = successive calls of the module: running statistics in call order
            mean = red[g, :, 0] / cnt
            var = (red[g, :, 1] / cnt - mean * mean).clamp_min(0.0)
            invstd = torch.rsqrt(var + eps)
            out[g] = torch.stack([ga * invstd, be - mean * ga * invstd, mean, invstd]).float()
            if rmean is not None:
                unb = var * (cnt / (cnt - 1.0)) if cnt > 1 else var
                rmean.mul_(1.0 - momentum).add_((momentum * mean).float())
                rvar.mul_(1.0 - momentum).add_((momentum * unb).float())
        if z_bound is not None:
            z_bound.copy_((out[:, 0].abs() * y_amax + out[:, 1].abs()).max().reshape(1))
        return
    splits = min(64, tiles_per_group // 64)  # two-stage reduction once a group has thousands of tile partials
    scratch = torch.empty(groups * splits * 2 * C, dtype=torch.float64, device=partial.device) if splits > 1 else None
    call("df_bn_finalize2", ptr(partial), tiles_per_group, groups, C, count, ptr(gamma), ptr(beta), eps, momentum,
         ptr(rmean), ptr(rvar), ptr(bn_ss), ptr(scratch), splits, ptr(y_amax), ptr(z_bound), stream())


def _elt(t: torch.Tensor) -> int:
    return 1 if t.dtype == torch.bfloat16 else 0


def bn_gelu_apply(y: torch.Tensor, bn_ss: torch.Tensor, imgs_per_group: int, z: DfImg):
    with timed("bn_gelu_apply", bytes=(y.element_size() + (2.0 if z.elt == 1 else 4.0)) * y.numel()):          # read y, write z
        if z.elt == 2:      # pre-split z: its bound (df_bn_finalize2) is an INPUT here
            call("df_bn_gelu_apply_t", ptr(y), _elt(y), ptr(bn_ss), imgs_per_group, z, ptr(z._amax), stream())
            return
        a = amax_slot(y.device) if (h2_active() and z.elt == 0) else None     # max |z| for the fp16x2 convolution that reads z
        call("df_bn_gelu_apply_t", ptr(y), _elt(y), ptr(bn_ss), imgs_per_group, z, ptr(a), stream())
        if a is not None:
            z._amax = a


def _pow2_blocks(rows_per_group: int, cap: int = 512) -> int:
    """largest power of two <= cap dividing rows_per_group with >= 16 rows per block"""
    n = 1
    while n * 2 <= cap and rows_per_group % (n * 2) == 0 and rows_per_group // (n * 2) >= 16:
        n *= 2
    return n


def bn_gelu_bwd(dz: DfImg, y: torch.Tensor, bn_ss: torch.Tensor, imgs_per_group: int, groups: int, gamma_grad: bool = True,
                frozen: bool = False, dy_dtype: torch.dtype = torch.float32, dy_h2: bool = False,
                y_amax: Optional[torch.Tensor] = None, partial_pre=None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """-> dy [n,h,w,C] (dy_dtype: float32, or bfloat16 in the bf16-storage mode), dgamma [C], dbeta [C], dbias [C].
    dy_h2 (with y_amax = max |y| from the forward's conv epilogue): dy is written pre-split (ops.h2_empty) with the scale of the
    bound the finalisation derives from max |dz|, max |y| and the statistics.
    frozen: eval-mode BatchNorm (running statistics are constants): the batch-statistic terms of the data gradient
    vanish (coef = 0), dgamma / dbeta keep their form, and the conv bias gradient is no longer cancelled."""
    dev = y.device
    C = dz.c
    rows_per_group = imgs_per_group * dz.h * dz.w
    nbg = _pow2_blocks(rows_per_group)
    nblk = nbg * groups
    gb = 2.0 if dz.elt else 4.0
    if partial_pre is not None:
        # the partial sums came out of the epilogue of the data gradient that produced dz (ops.conv2d bwd_bn): [groups * nbp][C][2]
        partial, nbp = partial_pre
        nbg_fin = nbp
    else:
        partial = _f32(nblk, C, 2, device=dev)
        nbg_fin = nbg
        with timed("bn_gelu_bwd_reduce", bytes=(gb + y.element_size()) * y.numel()):     # read dz, y
            call("df_bn_gelu_bwd_reduce_t", dz, ptr(y), _elt(y), ptr(bn_ss), imgs_per_group, ptr(partial), nblk, stream())
    dy_bound = None
    if dy_h2:
        assert y_amax is not None and dy_dtype == torch.float32 and dz.elt == 0 and y.dtype == torch.float32
        dy_bound = amax_slot(dev)
        dz_amax = amax_of(dz, dev)
    if SYNC is not None and not frozen:
        red = partial.view(groups, nbg_fin, C, 2).to(torch.float64).sum(1)          # [groups, C, (sum g, sum g * xhat)], this rank
        dbeta, dgamma = red[:, :, 0].sum(0).float(), red[:, :, 1].sum(0).float()
        glob = SYNC.sum(red.clone()) / (float(rows_per_group) * SYNC.world)
        coef = glob.permute(0, 2, 1).contiguous().float()                        # [groups, 2, C]
        if dy_h2:
            ss = bn_ss.view(groups, 4, C)
            xh = (y_amax + ss[:, 2].abs()) * ss[:, 3]
            dy_bound.copy_((ss[:, 0].abs() * (1.13 * dz_amax + coef[:, 0].abs() + xh * coef[:, 1].abs())).max().reshape(1))
    else:
        dgamma, dbeta = _f32(C, device=dev), _f32(C, device=dev)
        coef = _f32(groups, 2, C, device=dev)
        if dy_h2:
            call("df_bn_bwd_finalize2", ptr(partial), nbg_fin, groups, C, rows_per_group, ptr(dgamma), ptr(dbeta), ptr(coef), ptr(bn_ss),
                 ptr(dz_amax), ptr(y_amax), ptr(dy_bound), stream())
        else:
            call("df_bn_bwd_finalize", ptr(partial), nbg_fin, groups, C, rows_per_group, ptr(dgamma), ptr(dbeta), ptr(coef), stream())
        if frozen:
            coef.zero_()
    dy = h2_empty(y.shape, dev, dy_bound) if dy_h2 else torch.empty(y.shape, dtype=dy_dtype, device=dev)
    dbp = _f32(nblk, C, device=dev)
    with timed("bn_gelu_bwd_apply", bytes=(gb + y.element_size() + dy.element_size()) * y.numel()):     # read dz, y; write dy
        a = dy_bound if dy_h2 else amax_slot(dev) if (h2_active() and dy_dtype == torch.float32) else None     # max |dy| for the fp16x2 dgrad / wgrad
        call("df_bn_gelu_bwd_apply_t", dz, ptr(y), _elt(y), ptr(bn_ss), ptr(coef), imgs_per_group, ptr(dy), 2 if dy_h2 else _elt(dy), ptr(dbp), nblk,
             ptr(a), stream())
        if a is not None and not dy_h2:
            dy._df_amax = (a, ver(dy))          # _lib.img() hands it on to the descriptors made of this tensor
    dbias = _f32(C, device=dev)
    call("df_colsum_finalize", ptr(dbp), nblk, C, 1, ptr(dbias), 0, stream())
    return dy, dgamma, dbeta, dbias


def colsum(x: DfImg, device) -> torch.Tensor:
    rows = x.n * x.h * x.w
    nblk = max(1, min(1024, rows // 64))
    partial = _f32(nblk, x.c, device=device)
    call("df_colsum_partial", x, ptr(partial), nblk, stream())
    out = _f32(x.c, device=device)
    call("df_colsum_finalize", ptr(partial), nblk, x.c, 1, ptr(out), 0, stream())
    return out


def weight_transpose(w_ohwi: torch.Tensor) -> torch.Tensor:
    """[Cout,kh,kw,Cin] -> [Cin,kh,kw,Cout]"""
    if WPREP is not None:
        hit = WPREP.wt(w_ohwi)
        if hit is not None:
            return hit
    key = epoch = None
    if W_AMAX is None:      # outside a trainer step: the transpose of an unchanged weight tensor is kept (and with it the fp16 planes
        key, epoch = _cache_key(w_ohwi)      # _split_h2 keeps for THAT tensor) -- a backward made ~30 fresh copies that never hit
        hit = _WT_CACHE.get(key) if key is not None else None
        if hit is not None and hit[2] == epoch:
            return hit[0]
    co, kh, kw, ci = w_ohwi.shape
    wt = torch.empty((ci, kh, kw, co), dtype=torch.float32, device=w_ohwi.device)
    call("df_weight_transpose", ptr(w_ohwi), ptr(wt), co, kh * kw, ci, stream())
    if key is not None:
        old = _WT_CACHE.get(key)
        if old is not None:      # the superseded transpose's planes go with it
            _PLANE_CACHE.pop(_cache_key(old[0])[0], None)
        if len(_WT_CACHE) > 256:
            _WT_CACHE.clear()
        wt._df_keep = True
        _WT_CACHE[key] = (wt, w_ohwi, epoch)
    return wt


def _wgrad3_name(stride: int) -> str:
    ring = int(os.environ.get("DF_WGRAD_RING", "2"))   # mirrors df_conv2d_wgrad's dispatch for 3x3 kernels
    if stride == 2:
        return "wgrad3_ring_kernel<16,2,2>" if os.environ.get("DF_WGRAD_RING_S2", "1") != "0" else "wgrad_kernel<3,2,32>"
    return f"wgrad3_ring_kernel<32,{ring},1>" if ring in (2, 3) else "wgrad_dma_kernel<3,1,32>"


def conv2d_wgrad(x: DfImg, dy: DfImg, ks: int, stride: int, dw: torch.Tensor, ld_co: Optional[int] = None,
                 accumulate: bool = False, row_counts: Optional[torch.Tensor] = None, rows_per_seg: int = 0, dw_off: int = 0,
                 want_bias: bool = False) -> Optional[torch.Tensor]:
    """dw (float memory [Cout][taps][Cin] with row pitch ld_co, starting dw_off elements in) (+)= dy^T x.
    want_bias: also return the bias gradient sum_p dy[p, :] (fused: the kernel already stages every dy tile)."""
    dev = dw.device
    pre = x.elt == 2 and dy.elt == 2       # both operands pre-split (h2 images): wgrad3_h2p_kernel
    assert pre or (x.elt != 2 and dy.elt != 2), "a pre-split tensor meets an fp32 one in a weight gradient: unpack it (ops.h2_unpack) or keep both split"
    if pre:
        assert ks == 3 and stride == 1 and row_counts is None and call("df_conv2d_wgrad_h2p_ok", x, dy, ks, stride) == 1
    t16 = bool(x.elt == 1 and dy.elt == 1)
    # round 5: the 1x1 layers' fp32 tensors split in flight (wgrad1_h2_kernel) -- fp16x2 products instead of the fp32 MFMA
    fly = (h2_active() and _h2_on()) or (MFMA_BF16 and os.environ.get("DF_WGRAD_BF_FLY", "1") != "0")   # (bf16 mode: the one-plane forms)
    h2_1 = bool(not pre and not t16 and ks == 1 and stride == 1 and row_counts is None and fly
                and call("df_conv2d_wgrad1_h2_ok", x, dy) == 1)
    h2_s2 = bool(not pre and not t16 and ks == 3 and stride == 2 and row_counts is None and fly
                 and call("df_conv2d_wgrad_s2_h2_ok", x, dy) == 1)       # the stride-2 3x3 layers likewise (wgrad3s2_h2_kernel)
    splits = (call("df_conv2d_wgrad_h2p_splits", x, dy) if pre else call("df_conv2d_wgrad1_h2_splits", x, dy) if h2_1
              else call("df_conv2d_wgrad_s2_h2_splits", x, dy) if h2_s2 else call("df_conv2d_wgrad_splits", x, dy, ks, stride))
    taps = ks * ks
    ws = _f32(splits * dy.c * taps * x.c, device=dev)
    prof = PROFILER
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    bias_ws = _f32(splits, dy.c, device=dev) if want_bias else None
    x3 = (not pre and not t16 and not MFMA_BF16 and ks == 3 and stride == 1 and row_counts is None
          and call("df_conv2d_wgrad_x3_ok", x, dy, ks, stride) == 1)
    h2 = x3 and _h2_on()
    if pre:
        call("df_conv2d_wgrad_h2p", x, dy, ptr(x._amax), ptr(dy._amax), ks, stride, ks // 2, ptr(ws), splits, ptr(bias_ws), stream())
    elif t16:      # bf16-storage mode: both tensors bfloat16 in memory (transposing-read kernel, 3x3 stride 1 only)
        call("df_conv2d_wgrad_bf16", x, dy, ks, stride, ks // 2, ptr(ws), splits, ptr(bias_ws), stream())
    elif h2_1:
        call("df_conv2d_wgrad1_h2", x, dy, None if MFMA_BF16 else ptr(amax_of(x, dev)), None if MFMA_BF16 else ptr(amax_of(dy, dev)),
             ptr(ws), splits, ptr(bias_ws), stream())
    elif h2_s2:
        call("df_conv2d_wgrad_s2_h2", x, dy, None if MFMA_BF16 else ptr(amax_of(x, dev)), None if MFMA_BF16 else ptr(amax_of(dy, dev)),
             ptr(ws), splits, ptr(bias_ws), stream())
    elif h2:     # fp32 mode: fp32-accurate product from two scaled fp16 planes per operand (wgrad3_x3_kernel<2>)
        call("df_conv2d_wgrad_h2", x, dy, ptr(amax_of(x, dev)), ptr(amax_of(dy, dev)), ks, stride, ks // 2, ptr(ws), splits, ptr(bias_ws),
             stream())
    elif x3:     # fp32 mode: fp32-accurate product from three bf16 planes per operand (wgrad3_x3_kernel)
        call("df_conv2d_wgrad_x3", x, dy, ks, stride, ks // 2, ptr(ws), splits, ptr(bias_ws), stream())
    else:
        call("df_conv2d_wgrad_mp", x, dy, ks, stride, ks // 2, ptr(ws), splits, ptr(row_counts), rows_per_seg, ptr(bias_ws),
             int(MFMA_BF16), stream())
    if prof is not None:
        e1.record()
        name = ("wgrad3_h2p_kernel<4>" if pre else f"wgrad1_h2_kernel<{128 if dy.c % 128 == 0 else 64},{128 if x.c >= 128 else 64}>" if h2_1 else "wgrad3s2_h2_kernel" if h2_s2 else "wgrad3_tr_kernel<4>" if t16 else "wgrad3_x3_kernel<2>" if h2 else "wgrad3_x3_kernel<3>" if x3 else f"wgrad1x1_kernel<{128 if x.c >= 128 else 64}>" if ks == 1 and dy.c % 128 == 0
                else (_wgrad3_name(stride) if ks == 3 else f"wgrad_kernel<{ks},{stride},32>"))
        # ^ mirrors df_conv2d_wgrad's dispatch (DMA form for 3x3 stride 1)
        tag = f"wgrad {ks}x{ks} s{stride} {x.c}->{dy.c} @{dy.h}x{dy.w} x{dy.n} {'fbh'[x.elt]}{'fbh'[dy.elt]}"
        esz = lambda d: (4, 2, 4)[d.elt]
        prof.records.append((name + ("/bf16" if MFMA_BF16 else ""), 2.0 * dy.n * dy.h * dy.w * taps * x.c * dy.c, e0, e1, tag,
                             float(x.n * x.h * x.w * x.c * esz(x) + dy.n * dy.h * dy.w * dy.c * esz(dy))))
    if want_bias and os.environ.get("DF_FUSE_REDUCE", "1") != "0":   # weight + bias partials in one launch
        db = _f32(dy.c, device=dev)
        call("df_conv2d_wgrad_reduce_bias", ptr(ws), splits, dy.c, taps, x.c, dw.data_ptr() + 4 * dw_off,
             taps * x.c if ld_co is None else ld_co, int(accumulate), ptr(bias_ws), ptr(db), stream())
        return db
    call("df_conv2d_wgrad_reduce", ptr(ws), splits, dy.c, taps, x.c, dw.data_ptr() + 4 * dw_off,
         taps * x.c if ld_co is None else ld_co, int(accumulate), stream())
    if want_bias:
        db = _f32(dy.c, device=dev)
        call("df_colsum_finalize", ptr(bias_ws), splits, dy.c, 1, ptr(db), 0, stream())
        return db
    return None


def upsample2x(x: DfImg, y: DfImg, align_corners: bool):
    if y.elt == 2:      # the upsampled half of a pre-split concatenation: scale from the concatenation's bound (>= max |x|)
        call("df_upsample2x_h2", x, y, int(align_corners), ptr(y._amax), stream())
        return
    call("df_upsample2x", x, y, int(align_corners), stream())


def upsample2x_bwd(dy: DfImg, dx: DfImg, align_corners: bool):
    call("df_upsample2x_bwd", dy, dx, int(align_corners), stream())


def small_outer(a: torch.Tensor, lda: int, na: int, b: Optional[torch.Tensor], ldb: int, nb: int, counts: torch.Tensor,
                rows_per_seg: int, nseg: int, rows: int) -> torch.Tensor:
    nblk = max(1, min(1024, rows // 256))
    partial = _f32(nblk, na * nb, device=a.device)
    call("df_small_outer", ptr(a), lda, na, ptr(b), ldb, nb, ptr(counts), rows_per_seg, nseg, rows, ptr(partial), nblk,
         stream())
    out = _f32(na, nb, device=a.device)
    call("df_colsum_finalize", ptr(partial), nblk, na * nb, 1, ptr(out), 0, stream())
    return out
