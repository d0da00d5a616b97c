// Shared definitions of the BEV convolution kernels (conv.hip: forward / data gradients; conv_wgrad.hip: weight gradients).
#pragma once
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace {

constexpr int BK = 32;
constexpr int LDT = BK;  // LDS row pitch (floats); slots swizzled instead of padded

// ---- bf16-operand mode (mixed-precision training, BASELINE configs[4] "bf16 MFMA"): tensors stay fp32 in HBM and LDS;
// the MFMA operands are rounded to bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32) as the fragments are read, products are
// exact in the fp32 accumulator (v_mfma_f32_32x32x16_bf16: 16 k per instruction instead of 2).  What torch.autocast(bf16)
// computes for a conv -- operands in bf16, fp32 accumulation -- with the output kept in fp32.  Templates carry `BF`.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
// fp16x2 operands (NP = 2 forms of the x3 kernels): a tensor with max|x| <= amax is scaled by the power of two s that puts amax
// into [2^14, 2^15) and split as  x s = hi + lo / 2048,  hi = fp16(x s), lo = fp16((x s - hi) 2048)  -- 22 significant bits for
// every element down to 2^-29 amax (where hi leaves fp16's normal range; below that the absolute error is < 2^-39 amax).
__device__ __forceinline__ float df_h2_scale(float amax) {
  const int e = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 255u);     // amax in [2^(e-127), 2^(e-126))
  const int f = min(max(268 - e, 1), 254);                                     // s = 2^(141 - e), clamped to normal floats
  return __builtin_bit_cast(float, (unsigned)f << 23);
}
constexpr float H2_LO = 2048.f, H2_LO_INV = 1.f / 2048.f;
__device__ __forceinline__ void df_h2_split(const float (&v)[8], float s, f16x8_t& hi, f16x8_t& lo) {
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float t = v[k] * s;
    hi[k] = (_Float16)t;
    lo[k] = (_Float16)((t - (float)hi[k]) * H2_LO);
  }
}
__device__ __forceinline__ bf16x8_t pack_bf16(const f32x4 lo, const f32x4 hi) {
  bf16x8_t r;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    r[k] = (__bf16)lo[k];
    r[4 + k] = (__bf16)hi[k];
  }
  return r;
}
__device__ __forceinline__ bf16x8_t pack_bf16(const float (&v)[8]) {
  bf16x8_t r;
#pragma unroll
  for (int k = 0; k < 8; ++k) r[k] = (__bf16)v[k];
  return r;
}

struct ConvParams {
  df_img x, y;
  const float* w;
  const float* bias;
  const float* scale;
  const float* shift;
  float* stats;
  int ks, stride, pad, mode, epi, accumulate;
  int M, K, N, tiles_m, tiles_n, hw_y;
  // stride-2 dgrad: output pixels are tiled per parity class (y&1, x&1) so that a tile only visits the taps that
  // can reach it (1, 2, 2 or 4 of 9) instead of multiplying zeros.  cls_tiles = row tiles per class (0 = off).
  int cls_tiles;
  unsigned x_bytes, w_bytes, dshift;  // DMA path: buffer extents (bytes) and the base shift that keeps offsets >= 0
  unsigned y_bytes;                   // extent of y in bytes if it fits 32-bit buffer offsets (branch-free epilogue), else 0
  int dbg;  // ablation switch (env DF_CONV_DBG): 1 = no global loads after the prologue, 2 = also no LDS stores
  int bf16; // MFMA operands rounded to bf16 (fp32 tensors, fp32 accumulation)
  int stats_mul;  // statistics rows per tile (1; 2 when a 256-row tile fills the partial table sized for 128-row tiles: the second is 0)
  const float* amax_x;   // fp16x2 form (conv_halo_x3_kernel<.., NP = 2>): upper bounds of max|x| and max|w| (device scalars) that
  const float* amax_w;   // set the power-of-two scales of the two fp16 planes
  const void* w2p;       // conv_dma_kernel<.., H2, BP>: the weights PRE-SPLIT ([hi plane | lo plane] fp16 of w s_w, df_split_h2 / df_weight_prep), or null
  unsigned* amax_y;      // optional: the epilogue leaves max |y| there (bit pattern, atomic max) for an fp16x2 consumer of y
  const float* bound_y;  // y.elt == 2 (pre-split output): the bound of max |y| that defines the output's power-of-two scale
  // DF_EPI_BWD_STATS (round 4; data gradient only): y = dz is the gradient of a BatchNorm + GELU layer's OUTPUT.  bwd_y = that layer's
  // conv output (fp32, the geometry of y), bwd_ss = its (scale, shift, mean, invstd) per statistic group [groups][4][N]; the
  // epilogue leaves per tile and channel (sum g, sum g xhat), g = dz gelu'(bn(y)), in `stats` -- the partials of the BatchNorm
  // backward, which otherwise cost a separate pass over dz and y (bn_gelu_bwd_reduce_kernel)
  const float* bwd_y;
  const float* bwd_ss;
  int rot;  // conv_halo_x3_kernel: rotated tap-row order (see there); env DF_CONV_ROT=0 restores the plain order
};
constexpr int DF_EPI_BWD_STATS = 3;
__host__ __device__ inline bool epi_stats(int epi) { return epi == DF_EPI_STATS || epi == DF_EPI_BWD_STATS; }

// row m of the (possibly class-ordered) GEMM -> image, output y, output x
struct RowDecode {
  int hw, w, cls_mode, py, px, hh, wh;
  __device__ __forceinline__ void operator()(int m, int& n, int& oy, int& ox) const {
    if (!cls_mode) {
      n = m / hw;
      const int rem = m - n * hw;
      oy = rem / w;
      ox = rem - oy * w;
    } else {
      const int hwh = hh * wh;
      n = m / hwh;
      const int rem = m - n * hwh;
      const int yy = rem / wh;
      oy = 2 * yy + py;
      ox = 2 * (rem - yy * wh) + px;
    }
  }
};

constexpr unsigned DMA_BAD = 0xFFFFFFFFu - (8u << 20);  // + soffset (< 8 MB) never wraps, always out of range

// any16: the caller accepts bfloat16 elements too (then 16-byte alignment is 8 elements)
// (elt = 2, the pre-split fp16x2 "h2" layout of round 4, has the geometry of the fp32 tensor: 4 bytes per element)
bool img_ok(const df_img& d, bool any16 = false) {
  const int a = d.elt == 1 ? 8 : 4;
  return d.ptr && df_aligned16(d.ptr) && d.n > 0 && d.h > 0 && d.w > 0 && d.c > 0 && d.grp_size > 0 && (d.elt == 0 || (any16 && (d.elt == 1 || d.elt == 2))) &&
         (d.n % d.grp_size) == 0 && (d.ld % a) == 0 && (d.img_stride % a) == 0 && (d.grp_off % a) == 0;
}

}  // namespace
