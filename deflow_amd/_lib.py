"""ctypes binding of libdeflow_amd.so (include/deflow_amd.h).  No fallback: if the HIP library is missing
or a call is rejected, this raises."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DF_LIB") or os.path.join(_HERE, "libdeflow_amd.so")   # DF_LIB: an alternate build for A/B runs (tools/)

_ERR = {-1: "DF_E_SHAPE", -2: "DF_E_ALIGN", -3: "DF_E_ARG", -4: "DF_E_WORKSPACE"}


class DfImg(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("c", C.c_int32),
                ("ld", C.c_int32), ("grp_size", C.c_int32), ("img_stride", C.c_int64), ("grp_off", C.c_int64),
                ("elt", C.c_int32), ("reserved", C.c_int32)]   # elt: 0 = float32, 1 = bfloat16 (bf16-storage training), 2 = pre-split fp16x2 planes ("h2", round 4)


class DfGeom(C.Structure):
    _fields_ = [("vx", C.c_float), ("vy", C.c_float), ("vz", C.c_float),
                ("minx", C.c_float), ("miny", C.c_float), ("minz", C.c_float),
                ("offx", C.c_float), ("offy", C.c_float), ("offz", C.c_float),
                ("gx", C.c_int32), ("gy", C.c_int32), ("gz", C.c_int32)]


class DfGruWeights(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("w_off", "b_off", "w_zr", "b_zr", "w_q", "b_q", "w_1", "b_1", "w_2", "b_2")]


class DfGruWeightsT(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("wt_zr", "wt_q", "wt_1")]


P, I, L, F = C.c_void_p, C.c_int, C.c_int64, C.c_float

# name -> argtypes (restype int unless listed in _RESTYPE)
_SIGS = {
    "df_version": [],
    "df_pillar2_rows_per_band": [I, I],
    "df_pillar2_tile": [],
    "df_pillar2_hist": [P, I, I, DfGeom, I, P, P],
    "df_pillar2_scan": [P, I, I, I, P, P, P, P],
    "df_pillar2_scatter": [P, I, I, DfGeom, I, P, P, P, P, P, P, P, P, P, P, P, P],
    "df_pillar2_band": [P, P, P, P, P, I, DfGeom, I, I, P, P, I, I, DfImg, P, P, P, P, P, P],
    "df_pillar2_band_sp": [P, P, P, P, P, I, DfGeom, I, I, P, P, I, I, DfImg, P, P, P, P, P, P, P, P],
    "df_cell_sort_ws_bytes": [L],
    "df_cell_sort": [P, L, L, P, P, P, P],
    "df_pfn_bn_finalize": [P, I, I, P, P, P, F, F, P, P, P, P],
    "df_pfn_bn_finalize2": [P, I, I, P, P, P, F, F, P, P, P, P, DfGeom, P, P],
    "df_pfn_bwd_stats": [P, P, P, P, I, DfGeom, P, P, I, I, DfImg, P, I, P],
    "df_pfn_bwd_finalize": [P, I, I, P, P, P, I, P, P],
    "df_pfn_bwd_weights": [P, P, P, P, I, DfGeom, P, P, I, I, P, DfImg, P, I, P],
    "df_sparse_in_wgrad": [P, P, I, I, I, I, P, DfImg, P, I, P],
    "df_sparse_conv3x3": [P, P, I, DfImg, P, P, DfImg, I, P],
    "df_sparse_conv3x3_h2": [P, P, I, DfImg, P, P, P, P, DfImg, I, P],
    "df_sparse_conv3x3_bf16": [P, P, I, DfImg, P, P, DfImg, I, P],
    "df_sparse_wgrad3x3": [P, P, I, DfImg, DfImg, P, P, I, P],
    "df_sparse_wgrad3x3_x2": [P, P, I, DfImg, DfImg, P, P, I, P],
    "df_pillar_input_grad": [P, P, I, I, I, I, P, P, DfImg, P, DfImg, I, I, P],
    "df_conv2d": [DfImg, P, P, DfImg, I, I, I, I, I, P, P, P, I, P],
    "df_conv2d_mp": [DfImg, P, P, DfImg, I, I, I, I, I, P, P, P, I, I, P],
    "df_conv2d_wgrad_mp": [DfImg, DfImg, I, I, I, P, I, P, I, P, I, P],
    "df_conv2d_tile_m": [L, I],
    "df_conv2d_w16": [DfImg, P, P, DfImg, I, I, I, I, I, P, P, P, I, P],
    "df_conv2d_w16_ok": [DfImg, DfImg, I, I, I, I],
    "df_conv2d_x3": [DfImg, P, P, DfImg, I, I, I, I, I, P, P, P, I, P],
    "df_conv2d_x3_ok": [DfImg, DfImg, I, I, I, I],
    "df_split_bf16x3": [P, P, L, P],
    "df_split_bf16x2_rows": [P, P, L, I, P],
    "df_absmax": [DfImg, P, P],
    "df_split_h2": [P, P, P, L, P],
    "df_conv2d_h2": [DfImg, P, P, P, P, DfImg, I, I, I, I, I, P, P, P, I, P, P],
    "df_conv2d_amax": [DfImg, P, P, DfImg, I, I, I, I, I, P, P, P, I, P, P],
    "df_conv2d_h2f": [DfImg, P, P, P, P, DfImg, I, I, I, I, I, P, P, P, I, P, P],
    "df_conv2d_h2f_wp": [DfImg, P, P, P, P, P, DfImg, P, I, I, I, I, I, P, P, P, I, P, P],
    "df_conv2d_h2f_wp_up": [DfImg, P, P, P, P, P, DfImg, P, DfImg, I, P],
    "df_conv2d_h2p": [DfImg, P, P, P, P, DfImg, P, I, I, I, I, I, P, P, P, I, P, P],
    "df_conv2d_h2p_ok": [DfImg, DfImg, I, I, I, I],
    "df_conv2d_h2p_dgrad_bn": [DfImg, P, P, P, DfImg, P, P, P, P, P],
    "df_conv2d_yh2": [DfImg, P, P, P, P, DfImg, P, I, I, I, I, I, P, P, P, I, P],
    "df_conv2d_wgrad_h2p": [DfImg, DfImg, P, P, I, I, I, P, I, P, P],
    "df_conv2d_wgrad_h2p_ok": [DfImg, DfImg, I, I],
    "df_conv2d_wgrad_h2p_splits": [DfImg, DfImg],
    "df_bn_finalize2": [P, I, I, I, L, P, P, F, F, P, P, P, P, I, P, P, P],
    "df_bn_bwd_finalize2": [P, I, I, I, L, P, P, P, P, P, P, P, P],
    "df_upsample2x_h2": [DfImg, DfImg, I, P, P],
    "df_h2_pack": [DfImg, P, DfImg, P],
    "df_h2_unpack": [DfImg, P, DfImg, P],
    "df_rows_l1max": [P, I, I, P, I, P, P, P],
    "df_h2_bound": [P, P, P, P, P, F, P],
    "df_weight_prep": [P, P, I, I, P, P, P, P],
    "df_conv2d_variant": [L, L, I, I],
    "df_conv2d_last_dma": [],
    "df_conv2d_bf16": [DfImg, P, P, DfImg, I, I, I, I, P, P, I, P],
    "df_cast_bf16": [P, P, L, I, I, I, P],
    "df_upsample2x_bf16": [DfImg, DfImg, I, P],
    "df_bn_finalize": [P, I, I, I, L, P, P, F, F, P, P, P, P, I, P],
    "df_bn_gelu_apply": [P, P, I, DfImg, P],
    "df_bn_gelu_apply_t": [P, I, P, I, DfImg, P, P],
    "df_bn_gelu_bwd_reduce_t": [DfImg, P, I, P, I, P, I, P],
    "df_bn_gelu_bwd_apply_t": [DfImg, P, I, P, P, I, P, I, P, I, P, P],
    "df_conv2d_wgrad_bf16": [DfImg, DfImg, I, I, I, P, I, P, P],
    "df_conv2d_wgrad_x3": [DfImg, DfImg, I, I, I, P, I, P, P],
    "df_conv2d_wgrad_x3_ok": [DfImg, DfImg, I, I],
    "df_conv2d_wgrad_h2": [DfImg, DfImg, P, P, I, I, I, P, I, P, P],
    "df_conv2d_wgrad1_h2_ok": [DfImg, DfImg],
    "df_conv2d_wgrad1_h2_splits": [DfImg, DfImg],
    "df_conv2d_wgrad1_h2": [DfImg, DfImg, P, P, P, I, P, P],
    "df_conv2d_wgrad_s2_h2_ok": [DfImg, DfImg],
    "df_conv2d_wgrad_s2_h2_splits": [DfImg, DfImg],
    "df_conv2d_wgrad_s2_h2": [DfImg, DfImg, P, P, P, I, P, P],
    "df_bn_gelu_bwd_reduce": [DfImg, P, P, I, P, I, P],
    "df_bn_bwd_finalize": [P, I, I, I, L, P, P, P, P],
    "df_bn_gelu_bwd_apply": [DfImg, P, P, P, I, P, P, I, P],
    "df_colsum_partial": [DfImg, P, I, P],
    "df_colsum_finalize": [P, I, I, I, P, I, P],
    "df_colsum_stage": [P, I, I, I, P, P],
    "df_weight_transpose": [P, P, I, I, I, P],
    "df_conv2d_wgrad_splits": [DfImg, DfImg, I, I],
    "df_conv2d_wgrad": [DfImg, DfImg, I, I, I, P, I, P, I, P, P],
    "df_conv2d_wgrad_reduce": [P, I, I, I, I, P, L, I, P],
    "df_conv2d_wgrad_reduce_bias": [P, I, I, I, I, P, L, I, P, P, P],
    "df_upsample2x": [DfImg, DfImg, I, P],
    "df_upsample2x_bwd": [DfImg, DfImg, I, P],
    "df_gru_decoder_fwd": [DfImg, DfImg, P, P, P, I, I, I, DfGruWeights, P, P, P],
    "df_gru_decoder_fwd_mp": [DfImg, DfImg, P, P, P, I, I, I, DfGruWeights, P, P, I, P],
    "df_gru_decoder_bwd_mp": [P, P, P, I, I, I, DfGruWeights, DfGruWeightsT, P, P, P, P, P, P, I, P],
    "df_gru_wgrad_mp": [P, P, P, I, I, I, P, I, I, P],
    "df_gru_decoder_fwd_bf16": [DfImg, DfImg, P, P, P, I, I, I, P, P, P, P, P, P, P, P, P, P, P, P],
    "df_gru_decoder_bwd": [P, P, P, I, I, I, DfGruWeights, DfGruWeightsT, P, P, P, P, P, P, P],
    "df_gru_wgrad_splits": [],
    "df_gru_xtab": [DfGruWeights, P, P],
    "df_gru_lean_partial_width": [],
    "df_gru_lean_fwd": [DfImg, DfImg, P, P, P, I, I, I, DfGruWeights, P, P, P, I, P],
    "df_gru_lean_bwd": [P, P, P, I, I, I, DfGruWeights, DfGruWeightsT, P, P, P, P, P, P, I, P],
    "df_gru_lean_wgrad": [P, P, P, I, I, I, P, I, I, P],
    "df_gru_lean_head_wgrad": [P, P, P, I, I, P, I, P],
    "df_gru_lean_finalize": [P, DfGruWeights, P, P, P, P, P, P],
    "df_gru_head_wgrad": [P, P, P, P, I, I, P, I, P],
    "df_gru_wgrad": [P, P, P, I, I, I, P, I, P],
    "df_gather_bwd": [P, P, P, P, I, I, DfImg, DfImg, I, I, I, P],
    "df_gather_bwd_m": [P, P, P, P, I, I, DfImg, DfImg, I, I, P, P],
    "df_small_outer": [P, I, I, P, I, I, P, I, I, L, P, I, P],
    "df_linear_decoder_fwd": [DfImg, DfImg, P, P, P, I, I, P, P, P, P, P, P, P, P],
    "df_linear_decoder_bwd": [DfImg, DfImg, P, P, P, P, I, I, P, P, P, P, P, P, P, P, P, P, P, P],
    "df_ego_transform": [P, P, I, I, P, P, P],
    "df_deflow_loss_fwd": [P, P, P, I, I, P, I, P],
    "df_wloss_fwd": [P, P, P, I, I, I, P, P, I, P, I, P],
    "df_wloss_finalize": [P, I, I, P, P, P],
    "df_wloss_bwd": [P, P, P, I, I, I, P, P, I, P, P, F, P, I, P],
    "df_deflow_loss_finalize": [P, I, I, P, P, P],
    "df_deflow_loss_bwd": [P, P, P, I, I, P, P, F, P, I, P],
    "df_gather_gt": [P, P, P, P, I, I, P, I, P],
    "df_adam_step": [P, P, P, P, L, F, F, F, F, I, F, P],
    "df_adam_step_dev": [P, P, P, P, L, F, F, F, F, P, F, P],
}
_RESTYPE = {"df_cell_sort_ws_bytes": C.c_int64}
_RAW = {"df_pillar2_rows_per_band", "df_pillar2_tile", "df_version", "df_cell_sort_ws_bytes", "df_conv2d_tile_m", "df_conv2d_wgrad_splits", "df_conv2d_variant", "df_conv2d_w16_ok", "df_conv2d_x3_ok", "df_conv2d_wgrad_x3_ok", "df_conv2d_h2p_ok", "df_conv2d_wgrad_h2p_ok", "df_conv2d_wgrad_h2p_splits", "df_conv2d_wgrad1_h2_ok", "df_conv2d_wgrad1_h2_splits", "df_conv2d_wgrad_s2_h2_ok", "df_conv2d_wgrad_s2_h2_splits", "df_conv2d_last_dma", "df_gru_wgrad_splits", "df_gru_lean_partial_width"}  # return values, not status

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load the HIP library; raise loudly if it has not been built (python -m deflow_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: build the HIP kernels first (python -m deflow_amd.build). "
                           "deflow_amd has no CPU or eager fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in _SIGS.items():
        if not hasattr(lib, name):
            continue  # header/export consistency is checked by tests/test_abi.py
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = _RESTYPE.get(name, C.c_int)
    _lib = lib
    return lib


def call(name: str, *args):
    fn = getattr(load(), name)
    rc = fn(*args)
    if name in _RAW:
        return rc
    if rc != 0:
        raise RuntimeError(f"{name} failed: {_ERR.get(rc, 'hipError ' + str(rc))}")
    return 0


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _elt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return 0
    if t.dtype == torch.bfloat16:
        return 1
    raise TypeError(f"df_img: unsupported dtype {t.dtype}")


def ver(t: torch.Tensor) -> int:
    """write epoch of a tensor for the max |x| records: torch's version counter; tensors created under torch.inference_mode() do
    not track one (reading it raises) -- they never carry a record (-1 matches nothing: the consumer measures with df_absmax)"""
    return -1 if t.is_inference() else t._version


def img(t: torch.Tensor, c: Optional[int] = None, c_off: int = 0) -> DfImg:
    """Descriptor of an NHWC-shaped tensor [N,H,W,C'] (stride(3) == 1); optional channel slice [c_off, c_off+c).
    float32 or bfloat16 (the element type travels in the descriptor; only the bf16-storage entry points accept bfloat16)."""
    assert t.dim() == 4 and t.stride(3) == 1 and t.stride(1) == t.shape[2] * t.stride(2), (t.shape, t.stride())
    n, h, w, cc = t.shape
    c = cc - c_off if c is None else c
    h2 = getattr(t, "_df_h2", None)     # h2 tensor (ops.h2_empty): float32 storage holding [hi | lo] fp16 lines; _df_h2 = the bound that defines its scale
    if h2 is not None:
        assert c % 32 == 0 and c_off % 32 == 0 and t.stride(2) % 32 == 0, "h2 images are addressed in whole 32-channel chunks"
        d = DfImg(t.data_ptr() + 4 * c_off, n, h, w, c, t.stride(2), n, t.stride(0), 0, 2, 0)
        d._amax = h2
        return d
    d = DfImg(t.data_ptr() + t.element_size() * c_off, n, h, w, c, t.stride(2), n, t.stride(0), 0, _elt(t), 0)
    # max |t| measured by the kernel that wrote t (ops.bn_gelu_bwd, ops.conv2d): a bound for any view of it -- as long as nobody
    # has written t since (the kernels write through raw pointers and leave torch's version counter alone; an in-place torch op
    # bumps it, and the bound is dropped)
    rec = getattr(t, "_df_amax", None)
    if rec is not None and rec[1] == ver(t) and rec[1] >= 0:
        d._amax = rec[0]
    if c_off == 0 and c == cc:
        d._src = t                       # whole-tensor descriptor: a producer may leave the bound on the tensor (ops.conv2d)
    return d


def img_pair(t: torch.Tensor, c: int) -> DfImg:
    """A channel-concatenated buffer [B,H,W,2c] viewed as 2B images of c channels: image (g, b) = cloud g of sample b.
    This is how the shared encoder reads/writes torch.cat((pc0_x, pc1_x), dim=1) without a copy."""
    assert t.dim() == 4 and t.shape[3] == 2 * c and t.stride(3) == 1 and t.stride(1) == t.shape[2] * t.stride(2)
    n, h, w, _ = t.shape
    assert getattr(t, "_df_h2", None) is None
    d = DfImg(t.data_ptr(), 2 * n, h, w, c, t.stride(2), n, t.stride(0), c, _elt(t), 0)
    rec = getattr(t, "_df_amax", None)      # (the pair view covers the whole tensor: its bound applies)
    if rec is not None and rec[1] == ver(t) and rec[1] >= 0:
        d._amax = rec[0]
    d._src = t
    return d
