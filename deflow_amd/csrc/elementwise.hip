// HBM-bound companions of the convolutions: BatchNorm2d statistics / normalise+GELU (forward and
// backward), per-channel column sums, bilinear x2 resampling.  All NHWC, 16 B per lane, fully
// coalesced; reductions are two-stage (per-block fp32 partials -> fp64 finalize) and therefore
// deterministic (no float atomics).  Semantics follow torch.nn.BatchNorm2d / nn.GELU() /
// F.interpolate(mode="bilinear") as used by ConvWithNorms [REF decoder.py:202-220] and the UNet.
#include <cstdlib>

#include "common.h"

namespace {

// ---- element-typed 4-wide access (E = df_img.elt: 0 = float32, 1 = bfloat16; idx in ELEMENTS) -- bf16-storage training:
// the BatchNorm / GELU passes are pure HBM streams, half the bytes is half the time
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <int E>
__device__ __forceinline__ f32x4 ldx4(const void* base, int64_t idx) {
  if constexpr (E == 0) {
    return ld4(reinterpret_cast<const float*>(base) + idx);
  } else {
    const u32x2 v = *reinterpret_cast<const u32x2*>(reinterpret_cast<const unsigned short*>(base) + idx);
    f32x4 r;
    r[0] = __builtin_bit_cast(float, v[0] << 16);
    r[1] = __builtin_bit_cast(float, v[0] & 0xffff0000u);
    r[2] = __builtin_bit_cast(float, v[1] << 16);
    r[3] = __builtin_bit_cast(float, v[1] & 0xffff0000u);
    return r;
  }
}
template <int E>
__device__ __forceinline__ void stx4(void* base, int64_t idx, f32x4 v) {
  if constexpr (E == 0) {
    st4(reinterpret_cast<float*>(base) + idx, v);
  } else {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 a, b;
    a[0] = (__bf16)v[0]; a[1] = (__bf16)v[1]; b[0] = (__bf16)v[2]; b[1] = (__bf16)v[3];   // RNE (v_cvt_pk_bf16_f32)
    u32x2 w;
    w[0] = __builtin_bit_cast(unsigned, a);
    w[1] = __builtin_bit_cast(unsigned, b);
    *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned short*>(base) + idx) = w;
  }
}

// ---- "h2" images (df_img.elt = 2, round 4): a tensor PRE-SPLIT for the fp16x2 convolution / weight-gradient kernels
// (conv.hip: conv_halo_x3_kernel<.., XP>, wgrad3_h2p_kernel).  Geometry of the fp32 tensor (4 bytes per element, same ld / strides);
// per pixel and 32-channel chunk ONE 128-byte line [32 x fp16 hi | 32 x fp16 lo] with  x s = hi + lo / 2048,  s = the power of two
// that puts a BOUND of max |x| into [2^14, 2^15) (df_h2_scale: the bound is a device scalar the producer knows before it writes
// -- BatchNorm statistics, weight norms -- so nothing synchronises and no second pass is needed).  Any bound within ~2^10 of the
// true maximum keeps 22 significant bits for every element that matters (see conv.hip); looser ones degrade gracefully: the
// ABSOLUTE error stays below 2^-36 of the bound.  Requires whole chunks: c, ld, img_stride, grp_off % 32 == 0, 128-byte base.
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float df_h2_scale(float amax) {   // (conv.hip's)
  const int e = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 255u);
  const int f = min(max(268 - e, 1), 254);
  return __builtin_bit_cast(float, (unsigned)f << 23);
}
// idx = element index of 4 consecutive channels (idx % 4 == 0) inside the image whose base is 128-byte aligned
__device__ __forceinline__ void st_h2x4(void* base, int64_t idx, f32x4 v, float s) {
  f16x4_t hi, lo;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float t = v[k] * s;
    hi[k] = (_Float16)t;
    lo[k] = (_Float16)((t - (float)hi[k]) * 2048.f);
  }
  char* b = reinterpret_cast<char*>(base) + (idx & ~31ll) * 4 + (idx & 31) * 2;
  *reinterpret_cast<f16x4_t*>(b) = hi;
  *reinterpret_cast<f16x4_t*>(b + 64) = lo;
}
__device__ __forceinline__ f32x4 ld_h2x4(const void* base, int64_t idx, float inv_s) {
  const char* b = reinterpret_cast<const char*>(base) + (idx & ~31ll) * 4 + (idx & 31) * 2;
  const f16x4_t hi = *reinterpret_cast<const f16x4_t*>(b), lo = *reinterpret_cast<const f16x4_t*>(b + 64);
  f32x4 r;
#pragma unroll
  for (int k = 0; k < 4; ++k) r[k] = ((float)hi[k] + (float)lo[k] * (1.f / 2048.f)) * inv_s;
  return r;
}
__device__ __forceinline__ void df_atomic_amax(unsigned* slot, float v) {   // v >= 0: bit patterns order like the values
  const unsigned m = __builtin_bit_cast(unsigned, v);
  if (m > __atomic_load_n(slot, __ATOMIC_RELAXED)) atomicMax(slot, m);
}

// fp32 image -> h2 image (tests, and any producer without a fused form) and back
__global__ __launch_bounds__(256) void h2_pack_kernel(df_img x, df_img y, const float* __restrict__ bound, int64_t total4, int unpack) {
  const int C4 = x.c >> 2;
  const int hw = x.h * x.w;
  const int c4s = df_pow2_shift(C4), hws = df_pow2_shift(hw);
  const float s = df_h2_scale(*bound), inv_s = 1.f / s;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = df_udiv(i, C4, c4s);
    const int c = (int)(i - m * C4) * 4;
    const int n = (int)df_udiv(m, hw, hws), pix = (int)(m - (int64_t)n * hw);
    const int64_t xi = df_img_base(x, n) + (int64_t)pix * x.ld + c, yi = df_img_base(y, n) + (int64_t)pix * y.ld + c;
    if (unpack) st4(reinterpret_cast<float*>(y.ptr) + yi, ld_h2x4(x.ptr, xi, inv_s));
    else st_h2x4(y.ptr, yi, ld4(reinterpret_cast<const float*>(x.ptr) + xi), s);
  }
}

// max over rows of sum |w[row, :]| (and max |bias|): the a-priori bound  max |conv(x)| <= max|x| * max_row ||w_row||_1 + max |b|
// of a convolution output that is written pre-split.  One block per row; integer atomic max of the bit pattern (exact, order-free).
__global__ __launch_bounds__(256) void rows_l1max_kernel(const float* __restrict__ w, int row_len, const float* __restrict__ bias, int nbias,
                                                         unsigned* __restrict__ l1max, unsigned* __restrict__ bmax) {
  __shared__ float red[4];
  const float* r = w + (int64_t)blockIdx.x * row_len;
  float a = 0.f;
  for (int i = threadIdx.x; i < row_len; i += 256) a += fabsf(r[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    df_atomic_amax(l1max, (red[0] + red[1]) + (red[2] + red[3]));
    if (bias && bmax && blockIdx.x == 0) {
      float b = 0.f;
      for (int i = 0; i < nbias; ++i) b = fmaxf(b, fabsf(bias[i]));
      df_atomic_amax(bmax, b);
    }
  }
}

// ---- per-step weight preparation of ALL convolution layers in ONE launch (round 4; df_weight_prep) ---------------------------
// What the fp32 step did per convolution call: weight_transpose (31 launches per step), split_h2 of the weights and of their
// transpose (39), row L1 norms for the a-priori output bounds (~10).  The parameters change once per optimizer step, so all of
// it is a function of the parameter arena: one kernel walks a layer table and leaves, per layer,
//   wt   [Cin][taps][Cout] fp32       the transposed weights (operand of the data gradients)
//   w2   2 x [Cout][taps][Cin] fp16   [hi | lo] planes of w  s_w   (fp16x2 forward; 3x3 stride-1 layers)
//   wt2  2 x [Cin][taps][Cout] fp16   ... of the transpose         (fp16x2 data gradient)
//   l1w, l1wt                         max_row sum |w[row, :]|, max_row sum |wt[row, :]|  (a-priori bounds of pre-split outputs)
//   bmax                              max |bias|
// One workgroup per weight ROW (of w: blocks [0, Cout); of wt: blocks [Cout, Cout + Cin)), row sums by a fixed tree and integer
// atomic max: deterministic.  s_w = df_h2_scale(*w_amax) with w_amax = max |p| over the whole arena (one df_absmax before).
struct df_wprep_layer {
  int64_t w_off, b_off;          // element offsets of weight / bias in the parameter arena (b_off < 0: no bias)
  int64_t wt_off, w2_off, wt2_off;   // element offsets in the output buffer (wt: floats; w2 / wt2: in floats too, 2 x numel halfs = numel floats)
  int32_t cout, taps, cin, split;    // split: also write the fp16 planes
  int32_t blk0;                      // first workgroup of this layer
  int32_t pad;
};
__global__ __launch_bounds__(256) void weight_prep_kernel(const float* __restrict__ params, const df_wprep_layer* __restrict__ tab, int nlayers,
                                                          const float* __restrict__ w_amax, float* __restrict__ out,
                                                          unsigned* __restrict__ norms /* [nlayers][3]: l1w, l1wt, bmax */) {
  __shared__ float red[4];
  int L = 0;
  for (int i = 1; i < nlayers; ++i)
    if ((int)blockIdx.x >= tab[i].blk0) L = i;      // (a few dozen layers: a linear scan of scalar loads)
  const df_wprep_layer t = tab[L];
  const int r = (int)blockIdx.x - t.blk0;
  const float s = df_h2_scale(*w_amax);
  const float* w = params + t.w_off;
  const int64_t numel = (int64_t)t.cout * t.taps * t.cin;
  float a = 0.f;
  if (r < t.cout) {                 // a row of w: [taps][cin] contiguous
    const int len = t.taps * t.cin;
    const float* row = w + (int64_t)r * len;
    _Float16* hi = reinterpret_cast<_Float16*>(out + t.w2_off) + (int64_t)r * len;
    for (int i = threadIdx.x; i < len; i += 256) {
      const float v = row[i];
      a += fabsf(v);
      if (t.split) {
        const float x = v * s;
        const _Float16 h = (_Float16)x;
        hi[i] = h;
        hi[numel + i] = (_Float16)((x - (float)h) * 2048.f);
      }
    }
  } else {                          // a row of the transpose: wt[ci][tap][co] = w[co][tap][ci]
    const int ci = r - t.cout, len = t.taps * t.cout;
    float* wt = out + t.wt_off + (int64_t)ci * len;
    _Float16* hi = reinterpret_cast<_Float16*>(out + t.wt2_off) + (int64_t)ci * len;
    for (int i = threadIdx.x; i < len; i += 256) {
      const int tap = i / t.cout, co = i - tap * t.cout;
      const float v = w[((int64_t)co * t.taps + tap) * t.cin + ci];
      a += fabsf(v);
      wt[i] = v;
      if (t.split) {
        const float x = v * s;
        const _Float16 h = (_Float16)x;
        hi[i] = h;
        hi[numel + i] = (_Float16)((x - (float)h) * 2048.f);
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    df_atomic_amax(norms + 3 * L + (r < t.cout ? 0 : 1), (red[0] + red[1]) + (red[2] + red[3]));
    if (r == 0 && t.b_off >= 0) {
      float b = 0.f;
      for (int i = 0; i < t.cout; ++i) b = fmaxf(b, fabsf(params[t.b_off + i]));
      df_atomic_amax(norms + 3 * L + 2, b);
    }
  }
}

// out = max(other, a * l1 * slack + b)   (nulls: other = 0, l1 = 1, b = 0): the bound of a pre-split conv output / concatenation
__global__ void h2_bound_kernel(float* __restrict__ out, const float* a, const float* l1, const float* b, const float* other, float slack) {
  float v = *a * (l1 ? *l1 : 1.f) * slack + (b ? *b : 0.f);
  if (other) v = fmaxf(v, *other);
  *out = v;
}

// ------------------------------------------------------------------ BN finalize ---------
// Stage A (large layers): block (channel block, group, split) sums its range of per-tile partials in double and
// leaves [sum, sum of squares] per channel in scratch[g][split][2][C] -- the biggest layer has 32768 tiles per group,
// far too many for the C/32 blocks of the finalize kernel alone (it took 250 us there).
__global__ __launch_bounds__(1024) void bn_stats_split_kernel(const float* __restrict__ partial, int tiles_per_group,
                                                             int C, int splits, double* __restrict__ scratch) {
  __shared__ double red[2][32][32];
  const int cl = threadIdx.x & 31, tl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  const int g = blockIdx.y, sp = blockIdx.z;
  const int t0 = (int)((int64_t)tiles_per_group * sp / splits), t1 = (int)((int64_t)tiles_per_group * (sp + 1) / splits);
  double s1 = 0.0, s2 = 0.0;
  if (c < C) {
    for (int t = t0 + tl; t < t1; t += 32) {
      const float* q = partial + (((int64_t)g * tiles_per_group + t) * C + c) * 2;
      s1 += (double)q[0];
      s2 += (double)q[1];
    }
  }
  red[0][tl][cl] = s1;
  red[1][tl][cl] = s2;
  __syncthreads();
  if (tl == 0 && c < C) {
    for (int k = 1; k < 32; ++k) {
      s1 += red[0][k][cl];
      s2 += red[1][k][cl];
    }
    double* o = scratch + ((int64_t)g * splits + sp) * 2 * C;
    o[c] = s1;
    o[C + c] = s2;
  }
}

// PRE = false: partial is the conv epilogue's float [group][tile][C][2]; PRE = true: stage A's double
// [group][split][2][C] (tiles_per_group = splits).
template <bool PRE>
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const void* __restrict__ partial_, int tiles_per_group,
                                                          int groups, int C, double count, const float* gamma,
                                                          const float* beta, float eps, float momentum,
                                                          float* running_mean, float* running_var,
                                                          float* __restrict__ bn_ss, const float* __restrict__ y_amax,
                                                          unsigned* __restrict__ z_bound) {
  __shared__ double red[2][32][32];
  const int cl = threadIdx.x & 31, tl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  for (int g = 0; g < groups; ++g) {
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
      for (int t = tl; t < tiles_per_group; t += 32) {
        if (PRE) {
          const double* q = reinterpret_cast<const double*>(partial_) + ((int64_t)g * tiles_per_group + t) * 2 * C;
          s1 += q[c];
          s2 += q[C + c];
        } else {
          const float* q = reinterpret_cast<const float*>(partial_) + (((int64_t)g * tiles_per_group + t) * C + c) * 2;
          s1 += (double)q[0];
          s2 += (double)q[1];
        }
      }
    }
    red[0][tl][cl] = s1;
    red[1][tl][cl] = s2;
    __syncthreads();
    if (tl == 0 && c < C) {
      for (int k = 1; k < 32; ++k) {
        s1 += red[0][k][cl];
        s2 += red[1][k][cl];
      }
      const double mean = s1 / count;
      double var = s2 / count - mean * mean;
      if (var < 0.0) var = 0.0;
      const double invstd = 1.0 / sqrt(var + (double)eps);
      const double ga = gamma ? (double)gamma[c] : 1.0, be = beta ? (double)beta[c] : 0.0;
      float* o = bn_ss + (int64_t)g * 4 * C;
      o[0 * C + c] = (float)(ga * invstd);
      o[1 * C + c] = (float)(be - mean * ga * invstd);
      o[2 * C + c] = (float)mean;
      o[3 * C + c] = (float)invstd;
      // z = gelu(y scale + shift), |gelu(v)| <= |v|:  max |z| <= max_c |scale_c| max|y| + |shift_c|  -- the bound that scales the
      // pre-split z the apply pass writes next (y_amax: the conv epilogue's measurement, complete before this launch)
      if (z_bound) df_atomic_amax(z_bound, fabsf(o[0 * C + c]) * *y_amax + fabsf(o[1 * C + c]));
      if (running_mean) {
        const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mean);
        running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unb);
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ BN+GELU apply -------
template <int YE = 0, int ZE = 0>
__global__ __launch_bounds__(256) void bn_gelu_apply_kernel(const void* __restrict__ y, const float* __restrict__ bn_ss,
                                                            int imgs_per_group, df_img z, int64_t total4,
                                                            unsigned* __restrict__ amax) {
  const int C4 = z.c >> 2;
  const int hw = z.h * z.w;
  const int c4s = df_pow2_shift(C4), hws = df_pow2_shift(hw);
  float mf = 0.f;
  // ZE == 2: z is written pre-split (h2 image); `amax` is then an INPUT -- the bound of max |z| (df_bn_finalize2) that sets the scale
  float zs = 1.f;
  if constexpr (ZE == 2) zs = df_h2_scale(__builtin_bit_cast(float, *amax));
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = df_udiv(i, C4, c4s);
    const int c = (int)(i - m * C4) * 4;
    const int n = (int)df_udiv(m, hw, hws), pix = (int)(m - (int64_t)n * hw);
    const float* ss = bn_ss + (int64_t)(n / imgs_per_group) * 4 * z.c;
    const f32x4 v = ldx4<YE>(y, m * z.c + c), sc = ld4(ss + c), sh = ld4(ss + z.c + c);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = df_gelu(v[k] * sc[k] + sh[k]);
    if constexpr (ZE == 2) {
      st_h2x4(z.ptr, df_img_base(z, n) + (int64_t)pix * z.ld + c, o, zs);
    } else {
      stx4<ZE>(z.ptr, df_img_base(z, n) + (int64_t)pix * z.ld + c, o);
      mf = df_amax4(mf, o);
    }
  }
  if constexpr (ZE != 2)
    if (amax) df_block_amax(mf, amax);      // max |z| for the fp16x2 convolution that reads z next (uniform branch)
}

// the plane PRODUCER form with eight channels per thread (round 4; DF_BN_X8=0: four): one 16-byte store per plane instead of two
// 8-byte ones per four channels -- the h2 line of a 32-channel chunk is then written by four lanes x 2 stores instead of eight x 2
typedef _Float16 f16x8e_t __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void bn_gelu_apply8_kernel(const float* __restrict__ y, const float* __restrict__ bn_ss, int imgs_per_group,
                                                             df_img z, int64_t total8, const unsigned* __restrict__ bound) {
  const int C8 = z.c >> 3;
  const int hw = z.h * z.w;
  const int c8s = df_pow2_shift(C8), hws = df_pow2_shift(hw);
  const float zs = df_h2_scale(__builtin_bit_cast(float, *bound));
  // two element groups per iteration (round 6): four 16-byte loads in flight per thread before the first erf -- the pass ran at
  // 5.1 TB/s with two (DF_STREAM_UNROLL=1 at build time restores that)
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
#ifndef DF_STREAM_UNROLL
#define DF_STREAM_UNROLL 2
#endif
  for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < total8; i0 += DF_STREAM_UNROLL * stride) {
    f32x4 v0[DF_STREAM_UNROLL], v1[DF_STREAM_UNROLL];
    int64_t mm[DF_STREAM_UNROLL];
    int cc[DF_STREAM_UNROLL];
#pragma unroll
    for (int u = 0; u < DF_STREAM_UNROLL; ++u) {
      const int64_t i = i0 + u * stride;
      const int64_t ic = i < total8 ? i : i0;           // (a clamped, repeated load instead of a branch; its result is not stored)
      mm[u] = df_udiv(ic, C8, c8s);
      cc[u] = (int)(ic - mm[u] * C8) * 8;
      v0[u] = ld4(y + mm[u] * z.c + cc[u]);
      v1[u] = ld4(y + mm[u] * z.c + cc[u] + 4);
    }
#pragma unroll
    for (int u = 0; u < DF_STREAM_UNROLL; ++u) {
      if (i0 + u * stride >= total8) break;
      const int64_t m = mm[u];
      const int c = cc[u];
      const int n = (int)df_udiv(m, hw, hws), pix = (int)(m - (int64_t)n * hw);
      const float* ss = bn_ss + (int64_t)(n / imgs_per_group) * 4 * z.c;
      const f32x4 sc0 = ld4(ss + c), sc1 = ld4(ss + c + 4), sh0 = ld4(ss + z.c + c), sh1 = ld4(ss + z.c + c + 4);
      f16x8e_t hi, lo;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float t0 = df_gelu(v0[u][k] * sc0[k] + sh0[k]) * zs, t1 = df_gelu(v1[u][k] * sc1[k] + sh1[k]) * zs;
        hi[k] = (_Float16)t0;
        lo[k] = (_Float16)((t0 - (float)hi[k]) * 2048.f);
        hi[4 + k] = (_Float16)t1;
        lo[4 + k] = (_Float16)((t1 - (float)hi[4 + k]) * 2048.f);
      }
      const int64_t idx = df_img_base(z, n) + (int64_t)pix * z.ld + c;
      char* b = reinterpret_cast<char*>(z.ptr) + (idx & ~31ll) * 4 + (idx & 31) * 2;
      *reinterpret_cast<f16x8e_t*>(b) = hi;
      *reinterpret_cast<f16x8e_t*>(b + 64) = lo;
    }
  }
}

// block layout shared by the row-partitioned channel reductions: C/4 channel lanes x 256/(C/4) row lanes
struct RowPart {
  int c, row_lane, row_lanes;
};
__device__ __forceinline__ RowPart row_part(int C) {
  const int C4 = C >> 2;
  RowPart r;
  r.c = (threadIdx.x % C4) * 4;
  r.row_lane = threadIdx.x / C4;
  r.row_lanes = 256 / C4;
  return r;
}
// reduce NV float4 accumulators across the row lanes of the block; result valid on row_lane == 0
template <int NV>
__device__ __forceinline__ void block_reduce_rows(f32x4 (&acc)[NV], const RowPart& rp, float* lds /*[256*4*NV]*/) {
#pragma unroll
  for (int v = 0; v < NV; ++v) st4(lds + (v * 256 + threadIdx.x) * 4, acc[v]);
  __syncthreads();
  if (rp.row_lane == 0) {
    const int C4 = 256 / rp.row_lanes;
    for (int k = 1; k < rp.row_lanes; ++k)
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const f32x4 t = ld4(lds + (v * 256 + k * C4 + threadIdx.x) * 4);
        acc[v] += t;
      }
  }
}

// ------------------------------------------------------------------ BN+GELU backward ----
// pass 1: partial[blk][c][2] = sum over the block's rows of (dyh, dyh * xhat)
template <int GE = 0, int YE = 0>
__global__ __launch_bounds__(256) void bn_gelu_bwd_reduce_kernel(df_img dz, const void* __restrict__ y,
                                                                 const float* __restrict__ bn_ss, int imgs_per_group,
                                                                 float* __restrict__ partial, int64_t rows,
                                                                 int64_t rows_per_blk) {
  __shared__ __attribute__((aligned(16))) float lds[256 * 4 * 2];
  const int C = dz.c, hw = dz.h * dz.w;
  const int hws = df_pow2_shift(hw);
  const RowPart rp = row_part(C);
  const void* __restrict__ dzp = dz.ptr;
  const int64_t r_begin = (int64_t)blockIdx.x * rows_per_blk;
  const int64_t r_end = min(r_begin + rows_per_blk, rows);
  f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  if (r_begin < rows) {
    const int g = (int)(r_begin / hw) / imgs_per_group;
    const float* ss = bn_ss + (int64_t)g * 4 * C;
    const f32x4 sc = ld4(ss + rp.c), sh = ld4(ss + C + rp.c), mu = ld4(ss + 2 * C + rp.c), is = ld4(ss + 3 * C + rp.c);
    for (int64_t m = r_begin + rp.row_lane; m < r_end; m += rp.row_lanes) {
      const int n = (int)df_udiv(m, hw, hws), pix = (int)(m - (int64_t)n * hw);
      const f32x4 g4 = ldx4<GE>(dzp, df_img_base(dz, n) + (int64_t)pix * dz.ld + rp.c);
      const f32x4 yv = ldx4<YE>(y, m * C + rp.c);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float yh = yv[k] * sc[k] + sh[k];
        const float d = g4[k] * df_gelu_grad(yh);
        acc[0][k] += d;
        acc[1][k] += d * ((yv[k] - mu[k]) * is[k]);
      }
    }
  }
  block_reduce_rows<2>(acc, rp, lds);
  if (rp.row_lane == 0) {
    float* o = partial + ((int64_t)blockIdx.x * C + rp.c) * 2;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      o[k * 2 + 0] = acc[0][k];
      o[k * 2 + 1] = acc[1][k];
    }
  }
}

// 8 channels per workgroup x 128 slices of the partial rows: the data-gradient epilogues leave one partial row per workgroup (8192
// rows at 512 x 512 x 16), and a 32-channel workgroup walked them 256 deep with two to eight workgroups on the chip (77 us average,
// 250 us at 64 channels: 1.2 ms of the round-4 step).  Fixed summation order: bit-reproducible.
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int nblk_per_group,
                                                               int groups, int C, double count, float* dgamma,
                                                               float* dbeta, float* __restrict__ coef,
                                                               const float* __restrict__ bn_ss, const float* __restrict__ dz_amax,
                                                               const float* __restrict__ y_amax, unsigned* __restrict__ dy_bound) {
  __shared__ double red[2][128][8];
  const int cl = threadIdx.x & 7, tl = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + cl;
  double tg = 0.0, tb = 0.0;
  for (int g = 0; g < groups; ++g) {
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
      const float2* q = reinterpret_cast<const float2*>(partial) + ((int64_t)g * nblk_per_group) * C + c;
      int b = tl;
      for (; b + 384 < nblk_per_group; b += 512) {
        const float2 v0 = q[(int64_t)b * C], v1 = q[(int64_t)(b + 128) * C], v2 = q[(int64_t)(b + 256) * C], v3 = q[(int64_t)(b + 384) * C];
        s1 += ((double)v0.x + (double)v1.x) + ((double)v2.x + (double)v3.x);
        s2 += ((double)v0.y + (double)v1.y) + ((double)v2.y + (double)v3.y);
      }
      for (; b < nblk_per_group; b += 128) {
        const float2 v = q[(int64_t)b * C];
        s1 += (double)v.x;
        s2 += (double)v.y;
      }
    }
    red[0][tl][cl] = s1;
    red[1][tl][cl] = s2;
    __syncthreads();
    if (tl < 8) {
      s1 = 0.0, s2 = 0.0;
      for (int k = tl; k < 128; k += 8) {
        s1 += red[0][k][cl];
        s2 += red[1][k][cl];
      }
    }
    __syncthreads();
    if (tl < 8) {
      red[0][tl][cl] = s1;
      red[1][tl][cl] = s2;
    }
    __syncthreads();
    if (tl == 0 && c < C) {
      for (int k = 1; k < 8; ++k) {
        s1 += red[0][k][cl];
        s2 += red[1][k][cl];
      }
      coef[((int64_t)g * 2 + 0) * C + c] = (float)(s1 / count);
      coef[((int64_t)g * 2 + 1) * C + c] = (float)(s2 / count);
      if (dy_bound) {
        // dy = scale (dz gelu'(v) - c1 - xhat c2), |gelu'| <= 1.13, |xhat| <= (max|y| + |mean|) invstd: the bound that scales the
        // pre-split dy the apply pass writes next
        const float* ss = bn_ss + (int64_t)g * 4 * C;
        const float xh = (*y_amax + fabsf(ss[2 * C + c])) * ss[3 * C + c];
        df_atomic_amax(dy_bound, fabsf(ss[c]) * (1.13f * *dz_amax + fabsf((float)(s1 / count)) + xh * fabsf((float)(s2 / count))));
      }
      tb += s1;
      tg += s2;
    }
    __syncthreads();
  }
  if (tl == 0 && c < C) {
    if (dgamma) dgamma[c] = (float)tg;
    if (dbeta) dbeta[c] = (float)tb;
  }
}

// pass 2: dy = scale * (dyh - c1 - xhat * c2); dbias partial = column sums of dy
// DE = element type of dy: with bf16 the bias-gradient column sums are taken from the ROUNDED values (what the weight-gradient
// and data-gradient kernels will read)
template <int GE = 0, int YE = 0, int DE = 0>
__global__ __launch_bounds__(256) void bn_gelu_bwd_apply_kernel(df_img dz, const void* __restrict__ y,
                                                                const float* __restrict__ bn_ss,
                                                                const float* __restrict__ coef, int imgs_per_group,
                                                                void* __restrict__ dy, float* __restrict__ dbias_partial,
                                                                int64_t rows, int64_t rows_per_blk, unsigned* __restrict__ amax) {
  __shared__ __attribute__((aligned(16))) float lds[256 * 4];
  float mf = 0.f;
  // DE == 2: dy is written pre-split (a contiguous h2 image [rows][C]); `amax` is then an INPUT, the bound of max |dy| (df_bn_bwd_finalize2)
  float ds = 1.f;
  if constexpr (DE == 2) ds = df_h2_scale(__builtin_bit_cast(float, *amax));
  const int C = dz.c, hw = dz.h * dz.w;
  const int hws = df_pow2_shift(hw);
  const RowPart rp = row_part(C);
  const void* __restrict__ dzp = dz.ptr;
  const int64_t r_begin = (int64_t)blockIdx.x * rows_per_blk;
  const int64_t r_end = min(r_begin + rows_per_blk, rows);
  f32x4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
  if (r_begin < rows) {
    const int g = (int)(r_begin / hw) / imgs_per_group;
    const float* ss = bn_ss + (int64_t)g * 4 * C;
    const f32x4 sc = ld4(ss + rp.c), sh = ld4(ss + C + rp.c), mu = ld4(ss + 2 * C + rp.c), is = ld4(ss + 3 * C + rp.c);
    const f32x4 c1 = ld4(coef + ((int64_t)g * 2 + 0) * C + rp.c), c2 = ld4(coef + ((int64_t)g * 2 + 1) * C + rp.c);
    // two rows per iteration (round 6): both rows' loads go out before the first row's arithmetic; sums in the same order as before
    for (int64_t m0 = r_begin + rp.row_lane; m0 < r_end; m0 += 2 * rp.row_lanes) {
      f32x4 g4[2], yv[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int64_t m = m0 + u * rp.row_lanes < r_end ? m0 + u * rp.row_lanes : m0;     // (clamped repeat instead of a branch)
        const int n = (int)df_udiv(m, hw, hws), pix = (int)(m - (int64_t)n * hw);
        g4[u] = ldx4<GE>(dzp, df_img_base(dz, n) + (int64_t)pix * dz.ld + rp.c);
        yv[u] = ldx4<YE>(y, m * C + rp.c);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int64_t m = m0 + u * rp.row_lanes;
        if (m >= r_end) break;
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float yh = yv[u][k] * sc[k] + sh[k];
          const float d = g4[u][k] * df_gelu_grad(yh);
          const float xh = (yv[u][k] - mu[k]) * is[k];
          o[k] = sc[k] * (d - c1[k] - xh * c2[k]);
          if constexpr (DE == 1) o[k] = (float)(__bf16)o[k];
          acc[0][k] += o[k];
        }
        if constexpr (DE == 2) {
          st_h2x4(dy, m * C + rp.c, o, ds);
        } else {
          stx4<DE>(dy, m * C + rp.c, o);
          mf = df_amax4(mf, o);
        }
      }
    }
  }
  if constexpr (DE != 2)
    if (amax) df_block_amax(mf, amax);      // max |dy| for the fp16x2 data- / weight-gradient kernels (uniform branch)
  if (dbias_partial) {
    block_reduce_rows<1>(acc, rp, lds);
    if (rp.row_lane == 0) st4(dbias_partial + (int64_t)blockIdx.x * C + rp.c, acc[0]);
  }
}

// ------------------------------------------------------------------ column sums ---------
__global__ __launch_bounds__(256) void colsum_partial_kernel(df_img x, float* __restrict__ partial, int64_t rows,
                                                             int64_t rows_per_blk) {
  __shared__ __attribute__((aligned(16))) float lds[256 * 4];
  const int C = x.c, hw = x.h * x.w;
  const int hws = df_pow2_shift(hw);
  const RowPart rp = row_part(C);
  const float* __restrict__ xp = reinterpret_cast<const float*>(x.ptr);
  const int64_t r_begin = (int64_t)blockIdx.x * rows_per_blk;
  const int64_t r_end = min(r_begin + rows_per_blk, rows);
  f32x4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
  for (int64_t m = r_begin + rp.row_lane; m < r_end; m += rp.row_lanes) {
    const int n = (int)df_udiv(m, hw, hws), pix = (int)(m - (int64_t)n * hw);
    acc[0] += ld4(xp + df_img_base(x, n) + (int64_t)pix * x.ld + rp.c);
  }
  block_reduce_rows<1>(acc, rp, lds);
  if (rp.row_lane == 0) st4(partial + (int64_t)blockIdx.x * C + rp.c, acc[0]);
}

// gridDim.y > 1: row group blockIdx.y of the partials is summed into out[blockIdx.y][total] (first stage of a two-stage
// reduction when there are tens of thousands of partial rows)
__global__ __launch_bounds__(1024) void colsum_finalize_kernel(const float* __restrict__ partial, int nblk, int total,
                                                               float* out, int accumulate) {
  __shared__ double red[32][32];
  const int cl = threadIdx.x & 31, tl = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + cl;
  const int b0 = (int)((int64_t)nblk * blockIdx.y / gridDim.y), b1 = (int)((int64_t)nblk * (blockIdx.y + 1) / gridDim.y);
  out += (int64_t)blockIdx.y * total;
  double s = 0.0;
  if (i < total)
    for (int b = b0 + tl; b < b1; b += 32) s += (double)partial[(int64_t)b * total + i];
  red[tl][cl] = s;
  __syncthreads();
  if (tl == 0 && i < total) {
    for (int k = 1; k < 32; ++k) s += red[k][cl];
    out[i] = accumulate ? (float)((double)out[i] + s) : (float)s;
  }
}

// ------------------------------------------------------------------ bilinear x2 ---------
// PyTorch area_pixel_compute_source_index: align_corners ? dst*(in-1)/(out-1) : max((dst+0.5)*in/out-0.5, 0)
struct Lerp {
  int i0, i1;
  float l0, l1;
};
__device__ __forceinline__ Lerp lerp_src(int dst, int in, int out, int align_corners) {
  float src;
  if (align_corners) {
    const float sc = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
    src = sc * dst;
  } else {
    src = ((float)dst + 0.5f) * ((float)in / (float)out) - 0.5f;
    if (src < 0.f) src = 0.f;
  }
  Lerp r;
  r.i0 = (int)src;
  if (r.i0 > in - 1) r.i0 = in - 1;
  r.i1 = r.i0 + (r.i0 < in - 1 ? 1 : 0);
  r.l1 = src - (float)r.i0;
  r.l0 = 1.f - r.l1;
  return r;
}

// YE = 1: bfloat16 output (bf16-storage training: the upsampled half of an UpsampleSkip concatenation); the input stays fp32
template <int YE = 0>
__global__ __launch_bounds__(256) void upsample2x_kernel(df_img x, df_img y, int align_corners, int64_t total4, const float* __restrict__ y_bound) {
  float ys = 1.f;
  if constexpr (YE == 2) ys = df_h2_scale(*y_bound);      // pre-split output (the upsampled half of an h2 concatenation)
  const int C4 = y.c >> 2;
  const int c4s = df_pow2_shift(C4), yws = df_pow2_shift(y.w), yhs = df_pow2_shift(y.h);
  const float* __restrict__ xp = reinterpret_cast<const float*>(x.ptr);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t m = df_udiv(i, C4, c4s);
    const int c = (int)(i - m * C4) * 4;
    const int64_t mw = df_udiv(m, y.w, yws);
    const int X = (int)(m - mw * y.w);
    const int n = (int)df_udiv(mw, y.h, yhs), Y = (int)(mw - (int64_t)n * y.h);
    const Lerp ly = lerp_src(Y, x.h, y.h, align_corners), lx = lerp_src(X, x.w, y.w, align_corners);
    const float* b = xp + df_img_base(x, n) + c;
    const f32x4 v00 = ld4(b + ((int64_t)ly.i0 * x.w + lx.i0) * x.ld), v01 = ld4(b + ((int64_t)ly.i0 * x.w + lx.i1) * x.ld);
    const f32x4 v10 = ld4(b + ((int64_t)ly.i1 * x.w + lx.i0) * x.ld), v11 = ld4(b + ((int64_t)ly.i1 * x.w + lx.i1) * x.ld);
    const f32x4 o = ly.l0 * (lx.l0 * v00 + lx.l1 * v01) + ly.l1 * (lx.l0 * v10 + lx.l1 * v11);
    if constexpr (YE == 2) st_h2x4(y.ptr, df_img_base(y, n) + ((int64_t)Y * y.w + X) * y.ld + c, o, ys);
    else stx4<YE>(y.ptr, df_img_base(y, n) + ((int64_t)Y * y.w + X) * y.ld + c, o);
  }
}

// Eight channels per thread (round 6, second session): the x2 passes ran at 2.8-2.9 TB/s of their bytes -- half of what the BatchNorm +
// GELU plane producers reach -- with four channels per thread: two 8-byte stores per h2 plane and thread, and the index decode + the
// two source-index computations per 16 output bytes.  Same arithmetic per output value as upsample2x_kernel (bit-identical results).
template <int YE = 0>
__global__ __launch_bounds__(256) void upsample2x8_kernel(df_img x, df_img y, int align_corners, int64_t total8, const float* __restrict__ y_bound) {
  float ys = 1.f;
  if constexpr (YE == 2) ys = df_h2_scale(*y_bound);
  const int C8 = y.c >> 3;
  const int c8s = df_pow2_shift(C8), yws = df_pow2_shift(y.w), yhs = df_pow2_shift(y.h);
  const float* __restrict__ xp = reinterpret_cast<const float*>(x.ptr);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total8; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = df_udiv(i, C8, c8s);
    const int c = (int)(i - m * C8) * 8;
    const int64_t mw = df_udiv(m, y.w, yws);
    const int X = (int)(m - mw * y.w);
    const int n = (int)df_udiv(mw, y.h, yhs), Y = (int)(mw - (int64_t)n * y.h);
    const Lerp ly = lerp_src(Y, x.h, y.h, align_corners), lx = lerp_src(X, x.w, y.w, align_corners);
    const float* b = xp + df_img_base(x, n) + c;
    const float* p00 = b + ((int64_t)ly.i0 * x.w + lx.i0) * x.ld;
    const float* p01 = b + ((int64_t)ly.i0 * x.w + lx.i1) * x.ld;
    const float* p10 = b + ((int64_t)ly.i1 * x.w + lx.i0) * x.ld;
    const float* p11 = b + ((int64_t)ly.i1 * x.w + lx.i1) * x.ld;
    const f32x4 a00 = ld4(p00), a01 = ld4(p01), a10 = ld4(p10), a11 = ld4(p11);
    const f32x4 b00 = ld4(p00 + 4), b01 = ld4(p01 + 4), b10 = ld4(p10 + 4), b11 = ld4(p11 + 4);
    const f32x4 o0 = ly.l0 * (lx.l0 * a00 + lx.l1 * a01) + ly.l1 * (lx.l0 * a10 + lx.l1 * a11);
    const f32x4 o1 = ly.l0 * (lx.l0 * b00 + lx.l1 * b01) + ly.l1 * (lx.l0 * b10 + lx.l1 * b11);
    const int64_t idx = df_img_base(y, n) + ((int64_t)Y * y.w + X) * y.ld + c;
    if constexpr (YE == 2) {
      f16x8e_t hi, lo;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float t0 = o0[k] * ys, t1 = o1[k] * ys;
        hi[k] = (_Float16)t0;
        lo[k] = (_Float16)((t0 - (float)hi[k]) * 2048.f);
        hi[4 + k] = (_Float16)t1;
        lo[4 + k] = (_Float16)((t1 - (float)hi[4 + k]) * 2048.f);
      }
      char* q = reinterpret_cast<char*>(y.ptr) + (idx & ~31ll) * 4 + (idx & 31) * 2;
      *reinterpret_cast<f16x8e_t*>(q) = hi;
      *reinterpret_cast<f16x8e_t*>(q + 64) = lo;
    } else if constexpr (YE == 1) {
      typedef __bf16 bf16x8e_t __attribute__((ext_vector_type(8)));
      bf16x8e_t o;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        o[k] = (__bf16)o0[k];
        o[4 + k] = (__bf16)o1[k];
      }
      *reinterpret_cast<bf16x8e_t*>(reinterpret_cast<unsigned short*>(y.ptr) + idx) = o;
    } else {
      float* q = reinterpret_cast<float*>(y.ptr) + idx;
      st4(q, o0);
      st4(q + 4, o1);
    }
  }
}

// bf16 activations (inference path): the same PyTorch lerp semantics evaluated in fp32, 8 channels (16 bytes) per thread
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void upsample2x_bf16_kernel(df_img x, df_img y, int align_corners, int64_t total8) {
  const int C8 = y.c >> 3;
  const __bf16* __restrict__ xp = reinterpret_cast<const __bf16*>(x.ptr);
  __bf16* __restrict__ yp = reinterpret_cast<__bf16*>(y.ptr);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total8; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t m = i / C8;
    const int c = (int)(i - m * C8) * 8;
    const int X = (int)(m % y.w);
    m /= y.w;
    const int Y = (int)(m % y.h), n = (int)(m / y.h);
    const Lerp ly = lerp_src(Y, x.h, y.h, align_corners), lx = lerp_src(X, x.w, y.w, align_corners);
    const __bf16* b = xp + df_img_base(x, n) + c;
    const bf16x8_t v00 = *reinterpret_cast<const bf16x8_t*>(b + ((int64_t)ly.i0 * x.w + lx.i0) * x.ld);
    const bf16x8_t v01 = *reinterpret_cast<const bf16x8_t*>(b + ((int64_t)ly.i0 * x.w + lx.i1) * x.ld);
    const bf16x8_t v10 = *reinterpret_cast<const bf16x8_t*>(b + ((int64_t)ly.i1 * x.w + lx.i0) * x.ld);
    const bf16x8_t v11 = *reinterpret_cast<const bf16x8_t*>(b + ((int64_t)ly.i1 * x.w + lx.i1) * x.ld);
    bf16x8_t o;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      o[k] = (__bf16)(ly.l0 * (lx.l0 * (float)v00[k] + lx.l1 * (float)v01[k]) + ly.l1 * (lx.l0 * (float)v10[k] + lx.l1 * (float)v11[k]));
    *reinterpret_cast<bf16x8_t*>(yp + df_img_base(y, n) + ((int64_t)Y * y.w + X) * y.ld + c) = o;
  }
}

// gather form of the transpose: every input pixel sums the <= 6x6 output pixels that read it
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(df_img dy, df_img dx, int align_corners, int64_t total4) {
  const int C4 = dx.c >> 2;
  const int c4s = df_pow2_shift(C4), xws = df_pow2_shift(dx.w), xhs = df_pow2_shift(dx.h);
  const float* __restrict__ dyp = reinterpret_cast<const float*>(dy.ptr);
  float* __restrict__ dxp = reinterpret_cast<float*>(dx.ptr);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t m = df_udiv(i, C4, c4s);
    const int c = (int)(i - m * C4) * 4;
    const int64_t mw = df_udiv(m, dx.w, xws);
    const int xx = (int)(m - mw * dx.w);
    const int n = (int)df_udiv(mw, dx.h, xhs), yy = (int)(mw - (int64_t)n * dx.h);
    float wy[6], wx[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int Y = 2 * yy - 2 + k, X = 2 * xx - 2 + k;
      wy[k] = 0.f;
      wx[k] = 0.f;
      if (Y >= 0 && Y < dy.h) {
        const Lerp l = lerp_src(Y, dx.h, dy.h, align_corners);
        wy[k] = (l.i0 == yy ? l.l0 : 0.f) + (l.i1 == yy ? l.l1 : 0.f);
      }
      if (X >= 0 && X < dy.w) {
        const Lerp l = lerp_src(X, dx.w, dy.w, align_corners);
        wx[k] = (l.i0 == xx ? l.l0 : 0.f) + (l.i1 == xx ? l.l1 : 0.f);
      }
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* b = dyp + df_img_base(dy, n) + c;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      if (wy[a] == 0.f) continue;
      const int Y = 2 * yy - 2 + a;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        if (wx[k] == 0.f) continue;
        const int X = 2 * xx - 2 + k;
        acc += (wy[a] * wx[k]) * ld4(b + ((int64_t)Y * dy.w + X) * dy.ld);
      }
    }
    st4(dxp + df_img_base(dx, n) + ((int64_t)yy * dx.w + xx) * dx.ld + c, acc);
  }
}

// align_corners = false, eight channels per thread (round 6, second session): only the 4 x 4 outputs 2 y - 1 .. 2 y + 2 can read input
// row / column y; their weights come from the same source-index function as the forward (borders included), loads are unconditional at
// clamped addresses (a tap outside the image, or one that does not read this input, has weight 0), summation order = the gather form's
__global__ __launch_bounds__(256) void upsample2x_bwd8_kernel(df_img dy, df_img dx, int64_t total8) {
  const int C8 = dx.c >> 3;
  const int c8s = df_pow2_shift(C8), xws = df_pow2_shift(dx.w), xhs = df_pow2_shift(dx.h);
  const float* __restrict__ dyp = reinterpret_cast<const float*>(dy.ptr);
  float* __restrict__ dxp = reinterpret_cast<float*>(dx.ptr);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total8; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = df_udiv(i, C8, c8s);
    const int c = (int)(i - m * C8) * 8;
    const int64_t mw = df_udiv(m, dx.w, xws);
    const int xx = (int)(m - mw * dx.w);
    const int n = (int)df_udiv(mw, dx.h, xhs), yy = (int)(mw - (int64_t)n * dx.h);
    float wy[4], wx[4];
    int Yc[4], Xc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int Y = 2 * yy - 1 + k, X = 2 * xx - 1 + k;
      Yc[k] = min(max(Y, 0), dy.h - 1);
      Xc[k] = min(max(X, 0), dy.w - 1);
      const Lerp l = lerp_src(Yc[k], dx.h, dy.h, 0), r = lerp_src(Xc[k], dx.w, dy.w, 0);
      wy[k] = (Y == Yc[k]) ? (l.i0 == yy ? l.l0 : 0.f) + (l.i1 == yy ? l.l1 : 0.f) : 0.f;
      wx[k] = (X == Xc[k]) ? (r.i0 == xx ? r.l0 : 0.f) + (r.i1 == xx ? r.l1 : 0.f) : 0.f;
    }
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const float* b = dyp + df_img_base(dy, n) + c;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float* row = b + (int64_t)Yc[a] * dy.w * dy.ld;
      f32x4 v0[4], v1[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v0[k] = ld4(row + (int64_t)Xc[k] * dy.ld);
        v1[k] = ld4(row + (int64_t)Xc[k] * dy.ld + 4);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float w = wy[a] * wx[k];
        acc0 += w * v0[k];
        acc1 += w * v1[k];
      }
    }
    float* q = dxp + df_img_base(dx, n) + ((int64_t)yy * dx.w + xx) * dx.ld + c;
    st4(q, acc0);
    st4(q + 4, acc1);
  }
}

// any16: the entry point has a bfloat16 form (bf16 rows are accessed 8 bytes = 4 elements at a time)
bool img_ok(const df_img& d, bool any16 = false) {
  return d.ptr && df_aligned16(d.ptr) && d.n > 0 && d.h > 0 && d.w > 0 && d.c > 0 && (d.c % 4) == 0 && d.grp_size > 0 &&
         (d.elt == 0 || (any16 && d.elt == 1)) &&
         (d.n % d.grp_size) == 0 && (d.ld % 4) == 0 && (d.img_stride % 4) == 0 && (d.grp_off % 4) == 0;
}
// h2 image (elt = 2): whole 32-channel chunks at 128-byte lines
bool h2_ok(const df_img& d) {
  return d.elt == 2 && d.ptr && (((uintptr_t)d.ptr) & 127) == 0 && d.n > 0 && d.h > 0 && d.w > 0 && d.c > 0 && (d.c % 32) == 0 &&
         d.grp_size > 0 && (d.n % d.grp_size) == 0 && (d.ld % 32) == 0 && (d.img_stride % 32) == 0 && (d.grp_off % 32) == 0;
}
// DF_UP8=0: the four-channel forms of the bilinear x2 kernels (rounds 1-5) instead of the eight-channel ones
bool up8_on() {
  static const bool on = !(getenv("DF_UP8") && atoi(getenv("DF_UP8")) == 0);
  return on;
}
bool rowpart_ok(int C) { return C >= 4 && C <= 1024 && (C % 4) == 0 && (256 % (C / 4)) == 0; }
unsigned grid_for(int64_t total, int per_block = 256, unsigned cap = 256 * 16) {
  int64_t g = (total + per_block - 1) / per_block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (unsigned)g;
}

}  // namespace

extern "C" int df_bn_finalize(const float* partial, int tiles_per_group, int groups, int C, int64_t count_per_group,
                              const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                              float* running_var, float* bn_ss, double* scratch, int splits, void* stream) {
  return df_bn_finalize2(partial, tiles_per_group, groups, C, count_per_group, gamma, beta, eps, momentum, running_mean, running_var,
                         bn_ss, scratch, splits, nullptr, nullptr, stream);
}

extern "C" int df_bn_finalize2(const float* partial, int tiles_per_group, int groups, int C, int64_t count_per_group,
                               const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                               float* running_var, float* bn_ss, double* scratch, int splits, const float* y_amax, float* z_bound_,
                               void* stream) {
  unsigned* z_bound = reinterpret_cast<unsigned*>(z_bound_);
  DF_REQUIRE(!z_bound || y_amax, DF_E_ARG);
  DF_REQUIRE(partial && bn_ss && tiles_per_group > 0 && groups > 0 && C > 0 && count_per_group > 0, DF_E_ARG);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (scratch && splits > 1) {
    hipLaunchKernelGGL(bn_stats_split_kernel, dim3((C + 31) / 32, groups, splits), dim3(1024), 0, s, partial,
                       tiles_per_group, C, splits, scratch);
    DF_CHECK_LAUNCH();
    hipLaunchKernelGGL(bn_finalize_kernel<true>, dim3((C + 31) / 32), dim3(1024), 0, s, (const void*)scratch, splits,
                       groups, C, (double)count_per_group, gamma, beta, eps, momentum, running_mean, running_var, bn_ss, y_amax, z_bound);
  } else {
    hipLaunchKernelGGL(bn_finalize_kernel<false>, dim3((C + 31) / 32), dim3(1024), 0, s, (const void*)partial,
                       tiles_per_group, groups, C, (double)count_per_group, gamma, beta, eps, momentum, running_mean,
                       running_var, bn_ss, y_amax, z_bound);
  }
  DF_CHECK_LAUNCH();
  return DF_OK;
}

extern "C" int df_bn_gelu_apply_t(const void* y, int y_elt, const float* bn_ss, int imgs_per_group, df_img z, float* z_amax,
                                  void* stream) {
  unsigned* amax = reinterpret_cast<unsigned*>(z_amax);
  DF_REQUIRE(y && bn_ss && (img_ok(z, true) || (h2_ok(z) && y_elt == 0 && z_amax)) && df_aligned16(y) && imgs_per_group > 0 &&
                 (y_elt == 0 || y_elt == 1), DF_E_ARG);
  const int64_t total4 = (int64_t)z.n * z.h * z.w * (z.c / 4);
  const dim3 grid(grid_for(total4));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  static const int x8 = getenv("DF_BN_X8") ? atoi(getenv("DF_BN_X8")) : 1;
  if (z.elt == 2 && x8 && (z.c % 8) == 0)
    hipLaunchKernelGGL(bn_gelu_apply8_kernel, dim3(grid_for(total4 / 2)), dim3(256), 0, s, reinterpret_cast<const float*>(y), bn_ss, imgs_per_group, z,
                       total4 / 2, amax);
  else if (z.elt == 2) hipLaunchKernelGGL((bn_gelu_apply_kernel<0, 2>), grid, dim3(256), 0, s, y, bn_ss, imgs_per_group, z, total4, amax);
  else if (y_elt == 0 && z.elt == 0) hipLaunchKernelGGL((bn_gelu_apply_kernel<0, 0>), grid, dim3(256), 0, s, y, bn_ss, imgs_per_group, z, total4, amax);
  else if (y_elt == 1 && z.elt == 1) hipLaunchKernelGGL((bn_gelu_apply_kernel<1, 1>), grid, dim3(256), 0, s, y, bn_ss, imgs_per_group, z, total4, amax);
  else if (y_elt == 1) hipLaunchKernelGGL((bn_gelu_apply_kernel<1, 0>), grid, dim3(256), 0, s, y, bn_ss, imgs_per_group, z, total4, amax);
  else hipLaunchKernelGGL((bn_gelu_apply_kernel<0, 1>), grid, dim3(256), 0, s, y, bn_ss, imgs_per_group, z, total4, amax);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

extern "C" int df_bn_gelu_apply(const float* y, const float* bn_ss, int imgs_per_group, df_img z, void* stream) {
  DF_REQUIRE(z.elt == 0, DF_E_ARG);
  return df_bn_gelu_apply_t(y, 0, bn_ss, imgs_per_group, z, nullptr, stream);
}

extern "C" int df_bn_gelu_bwd_reduce_t(df_img dz, const void* y, int y_elt, const float* bn_ss, int imgs_per_group, float* partial,
                                       int nblk, void* stream) {
  DF_REQUIRE(img_ok(dz, true) && y && bn_ss && partial && nblk > 0 && rowpart_ok(dz.c) && (y_elt == 0 || y_elt == 1), DF_E_ARG);
  const int64_t rows = (int64_t)dz.n * dz.h * dz.w;
  const int64_t rows_per_group = (int64_t)imgs_per_group * dz.h * dz.w;
  DF_REQUIRE(rows % nblk == 0, DF_E_SHAPE);
  const int64_t rpb = rows / nblk;
  DF_REQUIRE(rows_per_group % rpb == 0, DF_E_SHAPE);  // a block never straddles two stat groups
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dz.elt == 0 && y_elt == 0) hipLaunchKernelGGL((bn_gelu_bwd_reduce_kernel<0, 0>), dim3(nblk), dim3(256), 0, s, dz, y, bn_ss, imgs_per_group, partial, rows, rpb);
  else if (dz.elt == 1 && y_elt == 1) hipLaunchKernelGGL((bn_gelu_bwd_reduce_kernel<1, 1>), dim3(nblk), dim3(256), 0, s, dz, y, bn_ss, imgs_per_group, partial, rows, rpb);
  else if (y_elt == 1) hipLaunchKernelGGL((bn_gelu_bwd_reduce_kernel<0, 1>), dim3(nblk), dim3(256), 0, s, dz, y, bn_ss, imgs_per_group, partial, rows, rpb);
  else hipLaunchKernelGGL((bn_gelu_bwd_reduce_kernel<1, 0>), dim3(nblk), dim3(256), 0, s, dz, y, bn_ss, imgs_per_group, partial, rows, rpb);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

extern "C" int df_bn_gelu_bwd_reduce(df_img dz, const float* y, const float* bn_ss, int imgs_per_group, float* partial,
                                     int nblk, void* stream) {
  DF_REQUIRE(dz.elt == 0, DF_E_ARG);
  return df_bn_gelu_bwd_reduce_t(dz, y, 0, bn_ss, imgs_per_group, partial, nblk, stream);
}

extern "C" int df_bn_bwd_finalize(const float* partial, int nblk_per_group, int groups, int C, int64_t count_per_group,
                                  float* dgamma, float* dbeta, float* coef, void* stream) {
  return df_bn_bwd_finalize2(partial, nblk_per_group, groups, C, count_per_group, dgamma, dbeta, coef, nullptr, nullptr, nullptr, nullptr,
                             stream);
}

extern "C" int df_bn_bwd_finalize2(const float* partial, int nblk_per_group, int groups, int C, int64_t count_per_group,
                                   float* dgamma, float* dbeta, float* coef, const float* bn_ss, const float* dz_amax,
                                   const float* y_amax, float* dy_bound, void* stream) {
  DF_REQUIRE(partial && (((uintptr_t)partial) & 7) == 0 && coef && nblk_per_group > 0 && groups > 0 &&
                 (!dy_bound || (bn_ss && dz_amax && y_amax)), DF_E_ARG);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 7) / 8), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream),
                     partial, nblk_per_group, groups, C, (double)count_per_group, dgamma, dbeta, coef, bn_ss, dz_amax, y_amax,
                     reinterpret_cast<unsigned*>(dy_bound));
  DF_CHECK_LAUNCH();
  return DF_OK;
}

extern "C" int df_bn_gelu_bwd_apply_t(df_img dz, const void* y, int y_elt, const float* bn_ss, const float* coef,
                                      int imgs_per_group, void* dy, int dy_elt, float* dbias_partial, int nblk, float* dy_amax,
                                      void* stream) {
  unsigned* amax = reinterpret_cast<unsigned*>(dy_amax);
  DF_REQUIRE(img_ok(dz, true) && y && bn_ss && coef && dy && nblk > 0 && rowpart_ok(dz.c) && (y_elt == 0 || y_elt == 1) &&
                 (dy_elt == 0 || dy_elt == 1 || (dy_elt == 2 && dz.elt == 0 && y_elt == 0 && dy_amax && (dz.c % 32) == 0 &&
                                                 (((uintptr_t)dy) & 127) == 0)), DF_E_ARG);
  const int64_t rows = (int64_t)dz.n * dz.h * dz.w;
  const int64_t rows_per_group = (int64_t)imgs_per_group * dz.h * dz.w;
  DF_REQUIRE(rows % nblk == 0, DF_E_SHAPE);
  const int64_t rpb = rows / nblk;
  DF_REQUIRE(rows_per_group % rpb == 0, DF_E_SHAPE);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define DF_BWD_APPLY(G, Y, D) \
  hipLaunchKernelGGL((bn_gelu_bwd_apply_kernel<G, Y, D>), dim3(nblk), dim3(256), 0, s, dz, y, bn_ss, coef, imgs_per_group, dy, dbias_partial, rows, rpb, amax)
  const int key = dy_elt == 2 ? 8 : dz.elt * 4 + y_elt * 2 + dy_elt;
  switch (key) {
    case 8: DF_BWD_APPLY(0, 0, 2); break;
    case 0: DF_BWD_APPLY(0, 0, 0); break;
    case 1: DF_BWD_APPLY(0, 0, 1); break;
    case 2: DF_BWD_APPLY(0, 1, 0); break;
    case 3: DF_BWD_APPLY(0, 1, 1); break;
    case 4: DF_BWD_APPLY(1, 0, 0); break;
    case 5: DF_BWD_APPLY(1, 0, 1); break;
    case 6: DF_BWD_APPLY(1, 1, 0); break;
    default: DF_BWD_APPLY(1, 1, 1); break;
  }
#undef DF_BWD_APPLY
  DF_CHECK_LAUNCH();
  return DF_OK;
}

extern "C" int df_bn_gelu_bwd_apply(df_img dz, const float* y, const float* bn_ss, const float* coef,
                                    int imgs_per_group, float* dy, float* dbias_partial, int nblk, void* stream) {
  DF_REQUIRE(dz.elt == 0, DF_E_ARG);
  return df_bn_gelu_bwd_apply_t(dz, y, 0, bn_ss, coef, imgs_per_group, dy, 0, dbias_partial, nblk, nullptr, stream);
}

extern "C" int df_colsum_partial(df_img x, float* partial, int nblk, void* stream) {
  DF_REQUIRE(img_ok(x) && partial && nblk > 0 && rowpart_ok(x.c), DF_E_ARG);
  const int64_t rows = (int64_t)x.n * x.h * x.w;
  const int64_t rpb = (rows + nblk - 1) / nblk;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(nblk), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, partial,
                     rows, rpb);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

extern "C" int df_colsum_finalize(const float* partial, int nblk, int C, int nvals, float* out, int accumulate,
                                  void* stream) {
  DF_REQUIRE(partial && out && nblk > 0 && C > 0 && nvals > 0, DF_E_ARG);
  const int total = C * nvals;
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3((total + 31) / 32), dim3(1024), 0,
                     reinterpret_cast<hipStream_t>(stream), partial, nblk, total, out, accumulate);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

extern "C" int df_colsum_stage(const float* partial, int nblk, int total, int groups, float* out, void* stream) {
  DF_REQUIRE(partial && out && nblk > 0 && total > 0 && groups > 0 && groups <= nblk, DF_E_ARG);
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3((total + 31) / 32, groups), dim3(1024), 0,
                     reinterpret_cast<hipStream_t>(stream), partial, nblk, total, out, 0);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

extern "C" int df_upsample2x(df_img x, df_img y, int align_corners, void* stream) {
  DF_REQUIRE(img_ok(x) && img_ok(y, true), DF_E_ARG);     // y may be bfloat16 (bf16-storage training)
  DF_REQUIRE(x.n == y.n && x.c == y.c && y.h == 2 * x.h && y.w == 2 * x.w, DF_E_SHAPE);
  const int64_t total4 = (int64_t)y.n * y.h * y.w * (y.c / 4);
  if (up8_on() && (y.c % 8) == 0 && (y.ld % 8) == 0 && (y.img_stride % 8) == 0 && (y.grp_off % 8) == 0) {
    if (y.elt) hipLaunchKernelGGL(upsample2x8_kernel<1>, dim3(grid_for(total4 / 2)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, y,
                                  align_corners, total4 / 2, nullptr);
    else hipLaunchKernelGGL(upsample2x8_kernel<0>, dim3(grid_for(total4 / 2)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, y,
                            align_corners, total4 / 2, nullptr);
    DF_CHECK_LAUNCH();
    return DF_OK;
  }
  if (y.elt) hipLaunchKernelGGL(upsample2x_kernel<1>, dim3(grid_for(total4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x,
                                y, align_corners, total4, nullptr);
  else hipLaunchKernelGGL(upsample2x_kernel<0>, dim3(grid_for(total4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x,
                          y, align_corners, total4, nullptr);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

// ---- pre-split ("h2") tensors, round 4 (layout: top of this file; consumers: df_conv2d_h2p, df_conv2d_wgrad_h2p) ----------------
// bilinear x2 into an h2 image (y.elt = 2), scale from *y_bound (>= max |x|: the outputs are convex combinations)
extern "C" int df_upsample2x_h2(df_img x, df_img y, int align_corners, const float* y_bound, void* stream) {
  DF_REQUIRE(img_ok(x) && h2_ok(y) && y_bound, DF_E_ARG);
  DF_REQUIRE(x.n == y.n && x.c == y.c && y.h == 2 * x.h && y.w == 2 * x.w, DF_E_SHAPE);
  const int64_t total4 = (int64_t)y.n * y.h * y.w * (y.c / 4);
  if (up8_on()) {     // (h2 images hold whole 32-channel chunks: c % 8 == 0)
    hipLaunchKernelGGL(upsample2x8_kernel<2>, dim3(grid_for(total4 / 2)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, y,
                       align_corners, total4 / 2, y_bound);
    DF_CHECK_LAUNCH();
    return DF_OK;
  }
  hipLaunchKernelGGL(upsample2x_kernel<2>, dim3(grid_for(total4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, y,
                     align_corners, total4, y_bound);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

// fp32 image -> h2 image with the scale of *bound (>= max |x|), and back (exact to 22 bits): tests / producers without a fused form
extern "C" int df_h2_pack(df_img x, const float* bound, df_img y, void* stream) {
  DF_REQUIRE(img_ok(x) && h2_ok(y) && bound && x.n == y.n && x.h == y.h && x.w == y.w && x.c == y.c, DF_E_ARG);
  const int64_t total4 = (int64_t)x.n * x.h * x.w * (x.c / 4);
  hipLaunchKernelGGL(h2_pack_kernel, dim3(grid_for(total4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, y, bound, total4, 0);
  DF_CHECK_LAUNCH();
  return DF_OK;
}
extern "C" int df_h2_unpack(df_img x, const float* bound, df_img y, void* stream) {
  DF_REQUIRE(h2_ok(x) && img_ok(y) && bound && x.n == y.n && x.h == y.h && x.w == y.w && x.c == y.c, DF_E_ARG);
  const int64_t total4 = (int64_t)x.n * x.h * x.w * (x.c / 4);
  hipLaunchKernelGGL(h2_pack_kernel, dim3(grid_for(total4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, y, bound, total4, 1);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

// *l1max <- max(*l1max, max_row sum |w[row, :]|), *bmax <- max(*bmax, max |bias|)  (slots zero-initialised by the caller; bias / bmax
// optional): the weight side of the a-priori bound of a pre-split convolution output
extern "C" int df_rows_l1max(const float* w, int rows, int row_len, const float* bias, int nbias, float* l1max, float* bmax, void* stream) {
  DF_REQUIRE(w && rows > 0 && row_len > 0 && l1max && (!bias || nbias > 0), DF_E_ARG);
  hipLaunchKernelGGL(rows_l1max_kernel, dim3(rows), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w, row_len, bias, nbias,
                     reinterpret_cast<unsigned*>(l1max), reinterpret_cast<unsigned*>(bmax));
  DF_CHECK_LAUNCH();
  return DF_OK;
}

// every convolution layer's per-step weight forms in one launch (kernel comment above): params = the parameter arena (or any
// base the table's offsets refer to), table = nlayers device records (df_wprep_layer, include/deflow_amd.h), total_blocks = sum over
// layers of (cout + cin), out = the output buffer the table's offsets address, norms = [nlayers][3] floats, zero-initialised
extern "C" int df_weight_prep(const float* params, const void* table, int nlayers, int total_blocks, const float* w_amax, float* out,
                              float* norms, void* stream) {
  DF_REQUIRE(params && table && nlayers > 0 && total_blocks > 0 && w_amax && out && norms, DF_E_ARG);
  hipLaunchKernelGGL(weight_prep_kernel, dim3(total_blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), params,
                     reinterpret_cast<const df_wprep_layer*>(table), nlayers, w_amax, out, reinterpret_cast<unsigned*>(norms));
  DF_CHECK_LAUNCH();
  return DF_OK;
}

// *out = max(other ? *other : 0, *a * (l1 ? *l1 : 1) * slack + (b ? *b : 0)): one thread; bounds of pre-split conv outputs / concatenations
extern "C" int df_h2_bound(float* out, const float* a, const float* l1, const float* b, const float* other, float slack, void* stream) {
  DF_REQUIRE(out && a && slack > 0.f, DF_E_ARG);
  hipLaunchKernelGGL(h2_bound_kernel, dim3(1), dim3(1), 0, reinterpret_cast<hipStream_t>(stream), out, a, l1, b, other, slack);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

extern "C" int df_upsample2x_bf16(df_img x, df_img y, int align_corners, void* stream) {
  DF_REQUIRE(x.ptr && y.ptr && df_aligned16(x.ptr) && df_aligned16(y.ptr), DF_E_ARG);
  DF_REQUIRE(x.n == y.n && x.c == y.c && y.h == 2 * x.h && y.w == 2 * x.w && (x.c % 8) == 0 && (x.ld % 8) == 0 && (y.ld % 8) == 0,
             DF_E_SHAPE);
  const int64_t total8 = (int64_t)y.n * y.h * y.w * (y.c / 8);
  hipLaunchKernelGGL(upsample2x_bf16_kernel, dim3(grid_for(total8)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x,
                     y, align_corners, total8);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

extern "C" int df_upsample2x_bwd(df_img dy, df_img dx, int align_corners, void* stream) {
  DF_REQUIRE(img_ok(dx) && img_ok(dy), DF_E_ARG);
  DF_REQUIRE(dx.n == dy.n && dx.c == dy.c && dy.h == 2 * dx.h && dy.w == 2 * dx.w, DF_E_SHAPE);
  const int64_t total4 = (int64_t)dx.n * dx.h * dx.w * (dx.c / 4);
  if (!align_corners && up8_on() && (dx.c % 8) == 0) {
    hipLaunchKernelGGL(upsample2x_bwd8_kernel, dim3(grid_for(total4 / 2)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dy, dx,
                       total4 / 2);
    DF_CHECK_LAUNCH();
    return DF_OK;
  }
  hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(grid_for(total4)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), dy, dx, align_corners, total4);
  DF_CHECK_LAUNCH();
  return DF_OK;
}
