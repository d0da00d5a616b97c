// Shared definitions of the BEV convolution kernels (conv.hip: forward / data gradients; conv_wgrad.hip: weight gradients).
#pragma once
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace dfconv {

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
  // stride-2 dgrad (cls_tiles > 0): tile -> parity class INTERLEAVED (class = tile & 3) instead of class-major (round 6, second session).
  // A class writes every other 256-byte pixel of every other row: with all workgroups of the chip on ONE class at a time the stores
  // (and the accumulate epilogue's loads) hit half of the memory channels; interleaved, the four classes of a region are in flight
  // together and their lines fill whole rows.  env DF_CONV_CLS_IL=0 restores the class-major order.
  int cls_il;
  // UPS (round 6, second session; conv_dma_kernel<.., UPS>, df_conv2d_h2f_wp_up): the 1x1 skip convolution of an UpsampleSkip block writes
  // the SECOND half (channels lat ..) of every pixel of the pre-split concatenation; with ups_on its workgroups also write the FIRST half of
  // their pixels -- the bilinear x2 of ups_t [N, H/2, W/2, lat] (fp32), the pass that was its own launch -- so that one kernel writes whole
  // 512-byte pixels instead of two kernels one half each (a half-pixel stride moves its bytes at 2.9 TB/s: r06_conv_experiments.txt 5a)
  df_img ups_t;
  int ups_on, ups_ac;
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
inline bool img_ok(const df_img& d, bool any16 = false) {
  const int a = d.elt == 1 ? 8 : 4;
  return d.ptr && df_aligned16(d.ptr) && d.n > 0 && d.h > 0 && d.w > 0 && d.c > 0 && d.grp_size > 0 && (d.elt == 0 || (any16 && (d.elt == 1 || d.elt == 2))) &&
         (d.n % d.grp_size) == 0 && (d.ld % a) == 0 && (d.img_stride % a) == 0 && (d.grp_off % a) == 0;
}

// BWS (compile time; only the pre-split-input 3x3 data-gradient instances carry it): the DF_EPI_BWD_STATS epilogue -- every other
// instance keeps the round-3 epilogue byte for byte (a run-time branch here cost the dominant kernels ~5 %: more scalar registers
// live across the main loop)
// PRE (round 6; conv_halo_x3p_kernel): acc already holds t = (y + bias) s, s = the plane output's scale (1 for an fp32 output): no bias
// add and no scaling here; the statistics and the maximum are taken of t and scaled back by 1 / s once per channel (a power of two
// commutes with every rounding of the sums -- bit-identical to the plain form)
template <int BM, int BN, int WM, int WN, bool BWS = false, bool PRE = false>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, f32x16 (&acc)[BM / WM / 32][BN / WN / 32], float* lds,
                                              const RowDecode& dec, int m0, int m_end, int n0, int tile_m, int tid_in = -1) {
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  // tid_in: the persistent kernel hands in an OPAQUE copy of threadIdx.x per tile -- otherwise the compiler hoists every lane-dependent
  // value of this epilogue out of its tile loop and carries ~100 registers through the main loop (measured: 65 spill stores per group)
  const int tid = tid_in >= 0 ? tid_in : (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, kh = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  // ---- epilogue ------------------------------------------------------------------------
  // LDS is free now: row -> output element offset table, then the stats scratch.
  int64_t* rowoff = reinterpret_cast<int64_t*>(lds);            // [BM]
  float* red = lds + 2 * BM;                                    // [WM][BN][2]
  if (tid < BM) {
    const int m = m0 + tid;
    int64_t off = -1;
    if (m < m_end) {
      int n, oy, ox;
      dec(m, n, oy, ox);
      off = df_img_base(p.y, n) + ((int64_t)oy * p.y.w + ox) * p.y.ld;
    }
    rowoff[tid] = off;
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (raw: the table is LDS traffic only)
  float* __restrict__ yp = reinterpret_cast<float*>(p.y.ptr);
  float amax_t = 0.f;                                           // max |y| over this thread's stored elements
  if (p.y_bytes) {
    // Straight-line stores: buffer stores whose offset is out of range for rows past the end (dropped by the hardware)
    // instead of a branch per element.  With the branch the compiler had to re-wait for the bias / scale loads inside every
    // element block -- s_waitcnt vmcnt(0), which also waits for the PREVIOUS element's store: 16 TM serialised write
    // round trips per wave.
    // Y16 (p.y.elt == 1, bf16-storage training): the value is rounded to bf16 (RNE) and stored as 2 bytes; the BatchNorm
    // statistics are taken from the ROUNDED values, i.e. of the tensor the normalisation pass will actually read.
    constexpr unsigned ROW_BAD = 0xFFFFFFFFu - (8u << 20);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y.ptr, 0, p.y_bytes, 0x00020000);
    // YH2 (p.y.elt == 2, round 4): y is written PRE-SPLIT for an fp16x2 consumer -- per pixel and 32-channel chunk one 128-byte line
    // [32 fp16 hi | 32 fp16 lo] of y s, s = df_h2_scale(*p.bound_y) (a bound of max |y| the caller knows before the launch).  A lane
    // owns one channel of the chunk (li): lanes exchange halves with their neighbour (one DPP move) so that every lane still stores
    // ONE dword per element -- even lanes the hi pair (channels li, li + 1), odd lanes the lo pair (li - 1, li).
    constexpr bool bws = BWS;
    const __amdgpu_buffer_rsrc_t y2r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bws ? p.bwd_y : reinterpret_cast<const float*>(p.y.ptr)), 0,
                                                                         p.y_bytes, 0x00020000);
    int bgrp = 0;
    if constexpr (bws) {
      int n_, oy_, ox_;
      dec(m0, n_, oy_, ox_);
      bgrp = n_ / p.y.grp_size;           // statistic group of this tile (a tile never straddles two)
    }
    // FULL (round 5): every row of the tile is inside the tensor (m0 + BM <= m_end -- all of this network's layers), so no element
    // needs its "row exists" test.  With the tests, the 64 per-element conditions of a wave (64-bit lane masks) did not fit the
    // scalar registers: they were spilled to VGPR lanes and every masked sum / max cost two v_readlane + a v_cndmask on top of its
    // arithmetic -- ~13 of the ~32 VALU instructions per output element of the fp16x2 epilogue (ISA count, conv_halo_x3p_kernel<512,64>),
    // while the epilogue is the part of a tile in which no wave of the workgroup issues MFMAs.
    auto body = [&](auto y_tag, auto full_tag) {
      constexpr int YT = decltype(y_tag)::value;
      constexpr bool FULL = decltype(full_tag)::value;
      constexpr bool Y16 = YT == 1, YH2 = YT == 2;
      constexpr int ESZ = Y16 ? 2 : 4;
      float sy = 1.f;
      if constexpr (YH2) sy = df_h2_scale(*p.bound_y);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int co = n0 + (wn * TN + j) * 32 + li;
        const float bia = p.bias ? p.bias[co] : 0.f;
        float sc = 1.f, sh = 0.f;
        if (p.epi == DF_EPI_BN_GELU) {
          sc = p.scale[co];
          sh = p.shift[co];
        }
        float s1 = 0.f, s2 = 0.f;
        float b_sc = 1.f, b_sh = 0.f, b_mu = 0.f, b_is = 1.f;
        if constexpr (bws) {
          const float* ss = p.bwd_ss + (int64_t)bgrp * 4 * p.N;
          b_sc = ss[co]; b_sh = ss[p.N + co]; b_mu = ss[2 * p.N + co]; b_is = ss[3 * p.N + co];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          unsigned ob[16];
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int64_t off = rowoff[(wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh];
            if constexpr (YH2) ob[e] = (FULL || off >= 0) ? (unsigned)((off + co - li) * 4 + (li & 1) * 64 + (li >> 1) * 4) : ROW_BAD;
            else ob[e] = (FULL || off >= 0) ? (unsigned)((off + co) * ESZ) : ROW_BAD;
          }
          float old[16];
          if constexpr (bws) {          // (fp32 y, no accumulation: checked by the launcher) the BatchNorm layer's conv output at the same elements
#pragma unroll
            for (int e = 0; e < 16; ++e) old[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(y2r, ob[e], 0, 0));
          }
          if (p.accumulate) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              if constexpr (Y16)
                old[e] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_raw_buffer_load_b16(yr, ob[e], 0, 0) << 16);
              else
                old[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yr, ob[e], 0, 0));
            }
          }
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            float v = PRE ? acc[i][j][e] : acc[i][j][e] + bia;
            if (!PRE && p.epi == DF_EPI_BN_GELU) v = df_gelu(v * sc + sh);
            if (p.accumulate) v += old[e];
            if constexpr (Y16) {
              const unsigned short h = __builtin_bit_cast(unsigned short, (__bf16)v);
              __builtin_amdgcn_raw_buffer_store_b16(h, yr, ob[e], 0, 0);
              v = __builtin_bit_cast(float, (unsigned)h << 16);
            } else if constexpr (YH2) {
              const float t = PRE ? v : v * sy;
              const _Float16 hi = (_Float16)t;
              const _Float16 lo = (_Float16)((t - (float)hi) * H2_LO);
              const unsigned hb = __builtin_bit_cast(unsigned short, hi), lb = __builtin_bit_cast(unsigned short, lo);
              const unsigned send = (li & 1) ? hb : lb;                 // odd lanes hand their hi to the even neighbour, even lanes their lo
              const unsigned recv = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send, 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, true);
              const unsigned word = (li & 1) ? (recv | (lb << 16)) : (hb | (recv << 16));
              __builtin_amdgcn_raw_buffer_store_b32(word, yr, ob[e], 0, 0);
            } else {
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yr, ob[e], 0, 0);
            }
            if (FULL || ob[e] != ROW_BAD) {
              if constexpr (bws) {
                const float g = v * df_gelu_grad(fmaf(old[e], b_sc, b_sh));
                s1 += g;
                s2 += g * ((old[e] - b_mu) * b_is);
              } else {
                s1 += v;
                s2 += v * v;
              }
              amax_t = fmaxf(amax_t, fabsf(v));
            }
          }
        }
        if (BWS || p.epi == DF_EPI_STATS) {
          if constexpr (PRE) {               // sums of t = y s -> sums of y (BWS: s2 = sum g xhat is linear in g, the plain one quadratic)
            const float isy = 1.f / sy;
            s1 *= isy;
            s2 *= BWS ? isy : isy * isy;
          }
          s1 += __shfl_xor(s1, 32);
          s2 += __shfl_xor(s2, 32);
          if (kh == 0) {
            const int cl = (wn * TN + j) * 32 + li;
            red[(wm * BN + cl) * 2 + 0] = s1;
            red[(wm * BN + cl) * 2 + 1] = s2;
          }
        }
      }
      if constexpr (PRE) amax_t *= 1.f / sy;
    };
    const bool full = m0 + BM <= m_end;      // (workgroup-uniform)
    if (full) {
      if (p.y.elt == 2) body(std::integral_constant<int, 2>{}, std::true_type{});
      else if (p.y.elt == 1) body(std::integral_constant<int, 1>{}, std::true_type{});
      else body(std::integral_constant<int, 0>{}, std::true_type{});
    } else {
      if (p.y.elt == 2) body(std::integral_constant<int, 2>{}, std::false_type{});
      else if (p.y.elt == 1) body(std::integral_constant<int, 1>{}, std::false_type{});
      else body(std::integral_constant<int, 0>{}, std::false_type{});
    }
  } else {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int co = n0 + (wn * TN + j) * 32 + li;
      const float bia = p.bias ? p.bias[co] : 0.f;
      float sc = 1.f, sh = 0.f;
      if (p.epi == DF_EPI_BN_GELU) {
        sc = p.scale[co];
        sh = p.shift[co];
      }
      float s1 = 0.f, s2 = 0.f;
  #pragma unroll
      for (int i = 0; i < TM; ++i) {
  #pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
          const int64_t off = rowoff[row];
          float v = acc[i][j][e] + bia;
          if (off >= 0) {
            if (p.epi == DF_EPI_BN_GELU) v = df_gelu(v * sc + sh);
            if (p.accumulate) v += yp[off + co];
            yp[off + co] = v;
            s1 += v;
            s2 += v * v;
            amax_t = fmaxf(amax_t, fabsf(v));
          }
        }
      }
      if (p.epi == DF_EPI_STATS) {
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (kh == 0) {
          const int cl = (wn * TN + j) * 32 + li;
          red[(wm * BN + cl) * 2 + 0] = s1;
          red[(wm * BN + cl) * 2 + 1] = s2;
        }
      }
    }
  }
  if (BWS || p.epi == DF_EPI_STATS) {
    // raw barrier: the partial sums are LDS traffic.  __syncthreads() also drains vmcnt -- every wave then waited for the acknowledgement
    // of the tile's global stores (1-2 us of a 36-45 us workgroup) before the last 64 threads could add up eight numbers
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (tid < BN) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) {
        s1 += red[(w * BN + tid) * 2 + 0];
        s2 += red[(w * BN + tid) * 2 + 1];
      }
      float* o = p.stats + ((int64_t)tile_m * p.stats_mul * p.N + n0 + tid) * 2;
      o[0] = s1;
      o[1] = s2;
      for (int r = 1; r < p.stats_mul; ++r) {
        o[(int64_t)r * p.N * 2] = 0.f;
        o[(int64_t)r * p.N * 2 + 1] = 0.f;
      }
    }
  }
  if (p.amax_y) df_block_amax(amax_t, p.amax_y);                // (uniform branch)
}

constexpr int LDH = 16;   // floats per row of a 16-bit LDS tile (32 channels of one pixel / one weight row: 64 bytes)

}  // namespace dfconv
using namespace dfconv;

// conv_x3p.hip (its own translation unit since round 6: the persistent pre-split 3x3 kernels -- 28 of the step's 70 ms -- compile in
// under a minute there instead of behind conv.hip's 3.5): tile form bm x bn = 256 x 128 (seg 1 / 2 / 4) or 512 x 64 (seg 1 / 2),
// bws = the DF_EPI_BWD_STATS epilogue.  p as prepared by conv2d_impl (conv.hip).
int df_launch_conv_halo_x3p(const dfconv::ConvParams& p, int bn, int seg, bool bws, hipStream_t s);
