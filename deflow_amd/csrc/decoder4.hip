// ConvGRU decoder, "lean" generation (round 5): forward + backward data pass + the small-gradient finalize
// ([REF decoder.py:123-199] and its derivative).  Two changes against decoder3.hip / decoder3_bwd.hip:
//
//  1. The offset encoding is affine in the 3-vector of point offsets: x = W_off o + b_off [REF decoder.py:172], and every
//     place x enters -- the x columns of the three gates [REF decoder.py:126-139] and of the head's first layer
//     [REF decoder.py:151,182] -- is linear in x.  So W[:, 128:] x + b = (W[:, 128:] W_off) o + (W[:, 128:] b_off + b) =: P o + c
//     with a [416][4] table (P | c) per optimizer step (df_gru_xtab: z, r, q rows, then the head's 32).  The kernels evaluate
//     the x contribution as three FMAs per gate value -- no x operand, no x GEMM chunks (a third of the K loop of every gate
//     GEMM in the un-hoisted form), no 96 hoisted registers.  Backwards the same identity removes every x-side GEMM:
//     with S_g[o][0..2] = sum over rows and iterations of dg_pre[row][o] * o[row][0..2] and S_g[o][3] = sum dg_pre[row][o]
//         dW_g[:, 128:] = S_g[:, :3] W_off^T + S_g[:, 3] b_off^T       d b_g = S_g[:, 3]
//         dW_off        = sum_g W_g[:, 128:]^T S_g[:, :3]              d b_off = sum_g W_g[:, 128:]^T S_g[:, 3]
//     (g over the three gates and the head layer): the backward kernel only accumulates the [416][4] sums S beside the
//     bias sums it took anyway, df_gru_lean_finalize does the two tiny products.  Only the fp32 summation order changes.
//
//  2. The forward saves the hidden state entering each iteration and h_T -- (T + 1) planes instead of 5 T + 1 -- and the
//     backward RECOMPUTES z, r, q from it (three 128 x 128 GEMMs per step on the matrix pipe it already drives), writing
//     the planes the weight-gradient pass streams (dz_pre, dr_pre, dq_pre, r * h) as before.  At the bench shape the three
//     kernels moved 44.6 GB per step through HBM (profiles/r04_pmc_hbm_bytes.txt); this form moves 5 + (5 + 16) + 20 planes.
//
// Layout, GEMM core and modes (0 fp32 MFMA, 1/2 bf16 operands, 3 bf16x2) are those of decoder3.hip / gemm_dma.h.
#include <cstdlib>

#include "common.h"
#include "gemm_dma.h"

// occupancy knobs of the forward kernel for the register-file experiment of tools/archive/pfn_race_probe10.sh (defaults = the shipped kernel)
#ifndef DF_GRU_LB
#define DF_GRU_LB 2
#endif
#ifdef DF_GRU_NUM_VGPR
#define DF_GRU_ATTR __attribute__((amdgpu_num_vgpr(DF_GRU_NUM_VGPR)))
#else
#define DF_GRU_ATTR
#endif

namespace {

using namespace gd;

constexpr int XT_ROWS = 416;     // z (128) | r (128) | q (128) | head layer 1 (32)
constexpr int PW4 = 1764;        // per-workgroup partial sums: S [416][4] | dW_2 [3][32] | d b_2 [3] | pad

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4 buf_ld4(rsrc_t r, unsigned voff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
}
__device__ __forceinline__ float buf_ld1(rsrc_t r, unsigned voff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0));
}
__device__ __forceinline__ f32x4 buf_ld4_bf16(rsrc_t r, unsigned voff) {   // 4 bf16 (8 bytes) -> 4 floats
  const u32x2 w = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, 0, 0));
  f32x4 v;
  v[0] = __builtin_bit_cast(float, w[0] << 16);
  v[1] = __builtin_bit_cast(float, w[0] & 0xffff0000u);
  v[2] = __builtin_bit_cast(float, w[1] << 16);
  v[3] = __builtin_bit_cast(float, w[1] & 0xffff0000u);
  return v;
}
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- the x table -----------------------------------------------------------------------------------------------------------
struct XtabParams {
  const float *w_off, *b_off, *w_zr, *b_zr, *w_q, *b_q, *w_1, *b_1;   // the fp32 parameters
  float* xtab;
};
__global__ __launch_bounds__(64) void gru_xtab_kernel(XtabParams p) {
  const int o = blockIdx.x * 64 + threadIdx.x;
  if (o >= XT_ROWS) return;
  const float* wrow;
  float bias;
  if (o < 256) { wrow = p.w_zr + o * 192 + 128; bias = p.b_zr[o]; }
  else if (o < 384) { wrow = p.w_q + (o - 256) * 192 + 128; bias = p.b_q[o - 256]; }
  else { wrow = p.w_1 + (o - 384) * 192 + 128; bias = p.b_1[o - 384]; }
  double a0 = 0., a1 = 0., a2 = 0., a3 = (double)bias;
  for (int j = 0; j < 64; ++j) {
    const double w = (double)wrow[j];
    a0 += w * (double)p.w_off[j * 3 + 0];
    a1 += w * (double)p.w_off[j * 3 + 1];
    a2 += w * (double)p.w_off[j * 3 + 2];
    a3 += w * (double)p.b_off[j];
  }
  st4(p.xtab + 4 * o, f32x4{(float)a0, (float)a1, (float)a2, (float)a3});
}

// ---- forward ----------------------------------------------------------------------------------------------------------------
struct Gru4Params {
  df_img before, after;
  const int32_t* coords;
  const float* offs;
  const int32_t* counts;
  int N, T;
  df_gru_weights w;      // w_zr, w_q, w_1 in the mode's form (fp32 / bf16 copies / bf16x2 rows); w_2, b_2 fp32
  const float* xtab;
  float* flow;
  float* hsave;          // [T + 1][B*N][128]: h entering iteration t, then h_T.  bf16 modes: planes 0 .. T-1 as bf16 half rows
  int64_t iter_stride;
};

// the wave's rows of (o_x, o_y, o_z, 0) -> its 16 x 4 LDS block; rows past the sample's count are zeros
__device__ __forceinline__ void offs_to_lds(float* Ow, const float* offs, int64_t grow0, unsigned nvalid, int lane) {
  if (lane < 16) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)lane < nvalid) {
      const float* o = offs + (grow0 + lane) * 3;
      v[0] = o[0]; v[1] = o[1]; v[2] = o[2];
    }
    st4(Ow + 4 * lane, v);
  }
}

// NWV (round 6) = waves per workgroup: 4 (64 points, two workgroups per CU: rounds 3-5) or 8 (128 points, ONE workgroup per CU).  The
// eight-wave form moves every weight chunk L2 -> LDS once per 128 points instead of once per 64: tools/bench_gru_lean.py with the DMA
// compiled out showed the weight stream -- its ISSUE, four 1-KB buffer_load ... lds per wave and chunk, not the wait for it: removing
// only the waits and barriers gained 4 % -- costing 20 % of the forward and 28 % of the backward.  Same arithmetic per point.
template <bool SAVE, int MODE, int NWV>
__global__ __launch_bounds__(64 * NWV, NWV == 4 ? DF_GRU_LB : NWV == 12 ? 3 : 2) DF_GRU_ATTR void gru_fwd4_kernel(Gru4Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr bool BF = MODE == 1 || MODE == 2, W16 = MODE == 2, X2 = MODE == 3;
  __shared__ __attribute__((aligned(16))) float Bs[2 * BT];           // 32 KB
  __shared__ __attribute__((aligned(16))) float As[NWV * 16 * LDH];   // 33.8 KB per four waves
  __shared__ __attribute__((aligned(16))) float Xt[XT_ROWS * 4];      // 6.6 KB
  __shared__ __attribute__((aligned(16))) float Os[NWV * 16 * 4];     // 1 KB per four waves
  const int b = blockIdx.y;
  const int cnt = p.counts[b];
  const int p0 = blockIdx.x * (16 * NWV);
  if (p0 >= cnt) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  float* Aw = As + wave * 16 * LDH;
  float* Ow = Os + wave * 64;
  const int wp0 = p0 + wave * 16;
  const int64_t grow0 = (int64_t)b * p.N + wp0;
  constexpr bool K8 = W16 || X2;
  constexpr int WSC = W16 ? 2 : 1;
  const float* a_lane = Aw + li * LDH + lq * (K8 ? 8 : 4);
  float* c_lane = Aw + 4 * lq * LDH + li;
  float* r_lane = Aw + (lane >> 5) * LDH + (lane & 31) * 4;
  const unsigned nvalid = (unsigned)min(max(cnt - wp0, 0), 16);
  const unsigned row_bytes = nvalid * 512u;
  const unsigned rl_off = ((lane >> 5) * 128 + (lane & 31) * 4) * 4;
  const float* w_z = p.w.w_zr;
  const float* w_r = p.w.w_zr + 128 * 192 / WSC;
  const float* w_q = p.w.w_q;

  WStreamT<MODE, NWV> ws;
  wstream_init(ws, Bs);
  dma_first<128, 192>(w_z, 0, Bs, ws);

  for (int i = tid; i < XT_ROWS; i += 64 * NWV) st4(Xt + 4 * i, ld4(p.xtab + 4 * i));
  offs_to_lds(Ow, p.offs, grow0, nvalid, lane);
  // ---- gather h0 = [before | after] -------------------------------------------------------------------------------
  {
    const float* bp = reinterpret_cast<const float*>(p.before.ptr) + df_img_base(p.before, b);
    const float* ap = reinterpret_cast<const float*>(p.after.ptr) + df_img_base(p.after, b);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int f = lane + 64 * j;
      const int pt = f >> 5, c4 = f & 31;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (wp0 + pt < cnt) {
        const int32_t* cc = p.coords + (grow0 + pt) * 3;
        const int64_t cell = (int64_t)cc[1] * p.before.w + cc[2];
        v = (c4 < 16) ? ld4(bp + cell * p.before.ld + c4 * 4) : ld4(ap + cell * p.after.ld + (c4 - 16) * 4);
      }
      st4(r_lane + 2 * j * LDH, v);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  f32x4 of[4];   // this lane's four rows of (o_x, o_y, o_z)  (12 waves: re-read where used -- 16 of the 168 registers)
#pragma unroll
  for (int r = 0; r < 4; ++r) of[r] = ld4(Ow + (4 * lq + r) * 4);
  // x contribution + bias of gate g (rows 128 g ..) as the initial accumulator
  auto xinit = [&](f32x4 (&acc)[8], int g) {
    if constexpr (NWV == 12) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        of[r] = ld4(Ow + (4 * lq + r) * 4);
        asm volatile("" : "+v"(of[r]));   // (a fresh value per use: not kept across the GEMMs)
      }
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const f32x4 tb = ld4(Xt + (g * 128 + 16 * t + li) * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t][r] = fmaf(tb[0], of[r][0], fmaf(tb[1], of[r][1], fmaf(tb[2], of[r][2], tb[3])));
    }
  };
  f32x4 xf[4];   // (unused operand slot of gemm<>: no GEMM here has an x part)
#pragma unroll
  for (int k = 0; k < 4; ++k) xf[k] = f32x4{0.f, 0.f, 0.f, 0.f};

  f32x4 h[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) h[t][r] = c_lane[r * LDH + 16 * t];

  auto save_rows = [&](int it) {   // coalesced copy of the wave's 16 x 128 A region (= h) to plane `it`
    const rsrc_t dst = make_rsrc(p.hsave + it * p.iter_stride + grow0 * 128, row_bytes);
    if (BF && it < p.T) {
      const unsigned ho = (lane >> 5) * 512 + (lane & 31) * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) buf_st4_bf16(dst, ho + j * 1024, ld4(r_lane + 2 * j * LDH));
      return;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) buf_st4(dst, rl_off + j * 1024, ld4(r_lane + 2 * j * LDH));
  };

  for (int it = 0; it < p.T; ++it) {
    if (SAVE) save_rows(it);
    f32x4 z[8], acc[8];
    xinit(z, 0);
    gemm<128, 4, false, 128>(w_z, 0, w_r, 0, a_lane, xf, ws, z);
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) z[t][r] = df_sigmoid_fast(z[t][r]);
    xinit(acc, 1);
    gemm<128, 4, false, 128>(w_r, 0, w_q, 0, a_lane, xf, ws, acc);
    // the wave's A region is private and the GEMM's closing barrier is behind us: overwrite h with r * h
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) c_lane[r * LDH + 16 * t] = df_sigmoid_fast(acc[t][r]) * h[t][r];
    wave_lds_sync();
    xinit(acc, 2);
    const bool last = it + 1 == p.T;
    if (!last) gemm<128, 4, false, 128>(w_q, 0, w_z, 0, a_lane, xf, ws, acc);
    else gemm<128, 4, false, 32>(w_q, 0, p.w.w_1, 0, a_lane, xf, ws, acc);
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float q = df_tanh_fast(acc[t][r]);
        h[t][r] = (1.f - z[t][r]) * h[t][r] + z[t][r] * q;
        c_lane[r * LDH + 16 * t] = h[t][r];
      }
    wave_lds_sync();
  }
  if (SAVE) save_rows(p.T);   // h_T
  // ---- MLP head: hid = W1[:, :128] h_T + (x table rows 384 ..) -----------------------------------------------------
  f32x4 hid[2];
  if constexpr (NWV == 12) {
#pragma unroll
    for (int r = 0; r < 4; ++r) of[r] = ld4(Ow + (4 * lq + r) * 4);
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const f32x4 tb = ld4(Xt + (384 + 16 * t + li) * 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) hid[t][r] = fmaf(tb[0], of[r][0], fmaf(tb[1], of[r][1], fmaf(tb[2], of[r][2], tb[3])));
  }
  gemm<32, 4, false, 32>(p.w.w_1, 0, nullptr, 0, a_lane, xf, ws, hid);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) c_lane[r * LDH + 16 * t] = df_gelu(hid[t][r]);
  wave_lds_sync();
  if (lane < 48) {
    const int pt = lane / 3, o = lane - pt * 3;
    if (wp0 + pt < cnt) {
      float a = p.w.b_2[o];
      for (int c = 0; c < 32; ++c) a = fmaf(p.w.w_2[o * 32 + c], Aw[pt * LDH + c], a);
      p.flow[(grow0 + pt) * 3 + o] = a;
    }
  }
#endif
}

// ---- backward ---------------------------------------------------------------------------------------------------------------
struct GruBwd4Params {
  const float* dflow;
  const float* offs;
  const int32_t* counts;
  int N, T;
  df_gru_weights w;        // forward-form GEMM weights (the recompute) + w_2
  df_gru_weights_t wt;     // transposed GEMM weights (the data gradients)
  const float* xtab;
  const float* hsave;      // [T + 1][B*N][128] from the forward
  float* gplanes;          // [4][T][B*N][128]: dz_pre | dr_pre | dq_pre | r * h  (bf16 modes: bf16 half rows)
  int64_t iter_stride, plane_stride;
  float* dh0;
  float* dpre1;
  float* partial;          // [blocks][PW4]
};

template <int MODE, int NWV>
__global__ __launch_bounds__(64 * NWV, 2) void gru_bwd4_kernel(GruBwd4Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr bool BF = MODE == 1 || MODE == 2, W16 = MODE == 2, X2 = MODE == 3;
  __shared__ __attribute__((aligned(16))) float Bs[2 * BT];
  __shared__ __attribute__((aligned(16))) float As[NWV * 16 * LDH];
  __shared__ __attribute__((aligned(16))) float Xt[XT_ROWS * 4];
  __shared__ __attribute__((aligned(16))) float Os[NWV * 16 * 4];
  constexpr int SMALL_W = 228;   // dW_2 [3][32] | d b_1 [32] | d b_2 [3] | pad | S_1[.][0..2] [3][32]
  __shared__ float Small[NWV * SMALL_W];
  const int b = blockIdx.y;
  const int cnt = p.counts[b];
  const int p0 = blockIdx.x * (16 * NWV);
  if (p0 >= cnt) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  float* Aw = As + wave * 16 * LDH;
  float* Ow = Os + wave * 64;
  const int wp0 = p0 + wave * 16;
  const int64_t grow0 = (int64_t)b * p.N + wp0;
  constexpr bool K8 = W16 || X2;
  constexpr int WSC = W16 ? 2 : 1;
  constexpr int CSC = K8 ? 2 : 1;
  const float* a_lane = Aw + li * LDH + lq * (K8 ? 8 : 4);
  float* c_lane = Aw + 4 * lq * LDH + li;
  float* r_lane = Aw + (lane >> 5) * LDH + (lane & 31) * 4;
  const unsigned nvalid = (unsigned)min(max(cnt - wp0, 0), 16);
  const unsigned row_bytes = nvalid * 512u;
  const unsigned rl_off = ((lane >> 5) * 128 + (lane & 31) * 4) * 4;
  const float* w_z = p.w.w_zr;
  const float* w_r = p.w.w_zr + 128 * 192 / WSC;
  const float* w_q = p.w.w_q;
  const float* wt_q = p.wt.wt_q;
  const float* wt_zr = p.wt.wt_zr;

  WStreamT<MODE, NWV> ws;
  wstream_init(ws, Bs);
  dma_first<32, 192>(p.w.w_1, 0, Bs, ws);

  auto lds_to_rows = [&](float* dst) {   // the wave's 16 x 128 A region -> global rows (coalesced; invalid rows dropped)
    const rsrc_t d = make_rsrc(dst + grow0 * 128, row_bytes);
#pragma unroll
    for (int j = 0; j < 8; ++j) buf_st4(d, rl_off + j * 1024, ld4(r_lane + 2 * j * LDH));
  };
  auto lds_to_plane = [&](float* dst) {  // ... into a plane of the weight-gradient pass (bf16 modes: bf16 half rows)
    if (!BF) { lds_to_rows(dst); return; }
    const rsrc_t d = make_rsrc(dst + grow0 * 128, row_bytes);
    const unsigned ho = (lane >> 5) * 512 + (lane & 31) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) buf_st4_bf16(d, ho + j * 1024, ld4(r_lane + 2 * j * LDH));
  };
  auto rows_to_lds = [&](const float* src, bool half) {   // saved rows -> the A region
    const rsrc_t s0 = make_rsrc(src + grow0 * 128, row_bytes);
    if (half) {
      const unsigned ho = (lane >> 5) * 512 + (lane & 31) * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) st4(r_lane + 2 * j * LDH, buf_ld4_bf16(s0, ho + j * 1024));
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) st4(r_lane + 2 * j * LDH, buf_ld4(s0, rl_off + j * 1024));
    }
  };
  auto lds_to_c = [&](f32x4 (&v)[8]) {
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[t][r] = c_lane[r * LDH + 16 * t];
  };
  auto c_to_lds = [&](const f32x4 (&v)[8]) {
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) c_lane[r * LDH + 16 * t] = v[t][r];
  };

  for (int i = tid; i < XT_ROWS; i += 64 * NWV) st4(Xt + 4 * i, ld4(p.xtab + 4 * i));
  offs_to_lds(Ow, p.offs, grow0, nvalid, lane);
  rows_to_lds(p.hsave + p.T * p.iter_stride, false);   // h_T
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // (the rows' offsets are re-read from LDS where they are used: 16 registers the recompute needs more)
  auto xinit = [&](f32x4 (&acc)[8], int g) {
    f32x4 of[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) of[r] = ld4(Ow + (4 * lq + r) * 4);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const f32x4 tb = ld4(Xt + (g * 128 + 16 * t + li) * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t][r] = fmaf(tb[0], of[r][0], fmaf(tb[1], of[r][1], fmaf(tb[2], of[r][2], tb[3])));
    }
  };
  f32x4 xf[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) xf[k] = f32x4{0.f, 0.f, 0.f, 0.f};

  // S_g of the three gates: column sums of the gate-gradient plane sitting in the A region, plain and weighted with the rows'
  // offsets (lane j owns columns j and j + 64)
  // Eight-wave form (round 6): five of the six S accumulators (z, r: both column halves; q: the first) live in wave-private LDS
  // ([gate][column half][lane] float4, 5 KB per wave: the one-workgroup-per-CU form has the room for exactly that) instead of 20 of
  // the 24 registers -- at 256 registers the four-wave form spills 16 and
  // reloads them inside the iteration loop (20 scratch instructions per iteration, each reload behind an s_waitcnt vmcnt(0) that
  // also drains the weight DMA in flight).
  constexpr bool SG_LDS = NWV == 8;
  __shared__ __attribute__((aligned(16))) float SgL[SG_LDS ? NWV * 5 * 64 * 4 : 4];
  float* sgl = SgL + (SG_LDS ? wave * (5 * 64 * 4) : 0);
  float sg[3][2][4];
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
      for (int d = 0; d < 4; ++d) sg[g][c][d] = 0.f;
      if (SG_LDS && g * 2 + c < 5) st4(sgl + ((g * 2 + c) * 64 + lane) * 4, f32x4{0.f, 0.f, 0.f, 0.f});
    }
  auto colsum = [&](int g) {
    float s[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll 4   // (fully unrolled the scheduler hoists all 48 LDS reads: 96 registers on top of four live planes)
    for (int r = 0; r < 16; ++r) {
      const f32x4 o = ld4(Ow + 4 * r);   // (broadcast read)
      const float v0 = Aw[r * LDH + lane], v1 = Aw[r * LDH + 64 + lane];
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        s[0][d] = fmaf(v0, o[d], s[0][d]);
        s[1][d] = fmaf(v1, o[d], s[1][d]);
      }
      s[0][3] += v0;
      s[1][3] += v1;
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      if (SG_LDS && g * 2 + c < 5) {
        float* q_ = sgl + ((g * 2 + c) * 64 + lane) * 4;
        const f32x4 o = ld4(q_);
        st4(q_, f32x4{o[0] + s[c][0], o[1] + s[c][1], o[2] + s[c][2], o[3] + s[c][3]});
      } else {
#pragma unroll
        for (int d = 0; d < 4; ++d) sg[g][c][d] += s[c][d];
      }
    }
  };

  // ---- MLP head backward -------------------------------------------------------------------------------------------
  f32x4 pre1[2];
  {
    f32x4 of[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) of[r] = ld4(Ow + (4 * lq + r) * 4);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const f32x4 tb = ld4(Xt + (384 + 16 * t + li) * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) pre1[t][r] = fmaf(tb[0], of[r][0], fmaf(tb[1], of[r][1], fmaf(tb[2], of[r][2], tb[3])));
    }
  }
  // pre1 = W1[:, :128] h_T + x part; the last chunk prefetches the first tile of W1^T (rows 0..127, 32 wide)
  gemm<32, 4, false, 128, 192, 32>(p.w.w_1, 0, p.wt.wt_1, 0, a_lane, xf, ws, pre1);
  {
    float df[4][3], sw2[2][3], sb1[2], s1[2][3], sdf[3];
    f32x4 of[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) of[r] = ld4(Ow + (4 * lq + r) * 4);
    const rsrc_t dfl = make_rsrc(p.dflow + grow0 * 3, nvalid * 12u);
    const rsrc_t dp1 = make_rsrc(p.dpre1 + grow0 * 32, nvalid * 128u);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int o = 0; o < 3; ++o) df[r][o] = buf_ld1(dfl, (4 * lq + r) * 12 + o * 4);
#pragma unroll
    for (int o = 0; o < 3; ++o) sdf[o] = df[0][o] + df[1][o] + df[2][o] + df[3][o];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int col = 16 * t + li;
      const float w20 = p.w.w_2[0 * 32 + col], w21 = p.w.w_2[1 * 32 + col], w22 = p.w.w_2[2 * 32 + col];
      sw2[t][0] = sw2[t][1] = sw2[t][2] = 0.f;
      s1[t][0] = s1[t][1] = s1[t][2] = 0.f;
      sb1[t] = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pre = pre1[t][r];
        const float dhid = df[r][0] * w20 + df[r][1] * w21 + df[r][2] * w22;
        const float dp = dhid * df_gelu_grad(pre);
        const float hv = df_gelu(pre);
        buf_st1(dp1, ((4 * lq + r) * 32 + col) * 4, dp);
#pragma unroll
        for (int o = 0; o < 3; ++o) sw2[t][o] += df[r][o] * hv;   // df is 0 on invalid rows
#pragma unroll
        for (int d = 0; d < 3; ++d) s1[t][d] = fmaf(dp, of[r][d], s1[t][d]);
        sb1[t] += dp;
        c_lane[r * LDH + 16 * t] = dp;                              // A operand of the next GEMM (k = 0..31)
      }
    }
    // the head's small sums leave the registers now: reduce over the 4 row groups and park them in LDS
    float* sm = Small + wave * SMALL_W;
    auto red4 = [&](float v) {
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      return v;
    };
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int o = 0; o < 3; ++o) {
        const float v = red4(sw2[t][o]), u = red4(s1[t][o]);
        if (lq == 0) {
          sm[o * 32 + 16 * t + li] = v;
          sm[132 + o * 32 + 16 * t + li] = u;
        }
      }
      const float v = red4(sb1[t]);
      if (lq == 0) sm[96 + 16 * t + li] = v;
    }
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      const float v = red4(sdf[o]);   // identical on the 16 lanes of a row group
      if (lane == 0) sm[128 + o] = v;
    }
  }
  wave_lds_sync();
  f32x4 dh[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) dh[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  gemm<128, 1, false, 128, 32, 192>(p.wt.wt_1, 0, w_z, 0, a_lane, xf, ws, dh);

  // ---- GRU steps in reverse ----------------------------------------------------------------------------------------
  for (int it = p.T - 1; it >= 0; --it) {
    float* pl_dz = p.gplanes + 0 * p.plane_stride + it * p.iter_stride;
    float* pl_dr = p.gplanes + 1 * p.plane_stride + it * p.iter_stride;
    float* pl_dq = p.gplanes + 2 * p.plane_stride + it * p.iter_stride;
    float* pl_rh = p.gplanes + 3 * p.plane_stride + it * p.iter_stride;
    f32x4 h[8], z[8], q[8], r[8];
    // (the last GEMM's closing barrier is behind every wave: the A region is free)
    rows_to_lds(p.hsave + it * p.iter_stride, BF);
    wave_lds_sync();
    // ---- recompute the gates [REF decoder.py:126-139] ----
    xinit(z, 0);
    gemm<128, 4, false, 128>(w_z, 0, w_r, 0, a_lane, xf, ws, z);
    xinit(r, 1);
    gemm<128, 4, false, 128>(w_r, 0, w_q, 0, a_lane, xf, ws, r);
    lds_to_c(h);
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        r[t][k] = df_sigmoid_fast(r[t][k]);
        c_lane[k * LDH + 16 * t] = r[t][k] * h[t][k];
      }
    wave_lds_sync();
    lds_to_plane(pl_rh);
    xinit(q, 2);
    gemm<128, 4, false, 128, 192, 256>(w_q, 0, wt_zr, 0, a_lane, xf, ws, q);
    // ---- h' = (1 - z) h + z q differentiated ----
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float d = dh[t][k], zz = df_sigmoid_fast(z[t][k]), qq = df_tanh_fast(q[t][k]);
        dh[t][k] = d * (1.f - zz);
        q[t][k] = d * zz * (1.f - qq * qq);                // dq_pre
        z[t][k] = d * (qq - h[t][k]) * zz * (1.f - zz);    // dz_pre
      }
    c_to_lds(z);
    wave_lds_sync();
    lds_to_plane(pl_dz);
    colsum(0);
    gemm<128, 4, false, 128, 256, 128>(wt_zr, 0, wt_q, 0, a_lane, xf, ws, dh);
    c_to_lds(q);
    wave_lds_sync();
    lds_to_plane(pl_dq);
    colsum(2);
    {
      f32x4 drh[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) drh[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      gemm<128, 4, false, 128, 128, 256>(wt_q, 0, wt_zr + 128 / CSC, 0, a_lane, xf, ws, drh);
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float d = drh[t][k], rr = r[t][k];
          dh[t][k] += d * rr;
          q[t][k] = d * h[t][k] * rr * (1.f - rr);   // dr_pre
        }
    }
    c_to_lds(q);
    wave_lds_sync();
    lds_to_plane(pl_dr);
    colsum(1);
    if (it > 0) gemm<128, 4, false, 128, 256, 192>(wt_zr + 128 / CSC, 0, w_z, 0, a_lane, xf, ws, dh);
    else gemm<128, 4, false, 128, 256, 192>(wt_zr + 128 / CSC, 0, nullptr, 0, a_lane, xf, ws, dh);
  }
  // ---- output: dh0 [rows,128] ----------------------------------------------------------------------------------------
  c_to_lds(dh);
  wave_lds_sync();
  lds_to_rows(p.dh0);
  // ---- per-workgroup partial sums (rows beyond cnt contributed exact zeros) --------------------------------------------
  // [NWV waves][PW4]: four waves fit the idle weight buffers; eight take the A regions (every wave's dh0 rows have left them first)
  float* red = NWV == 4 ? Bs : As;
  static_assert(NWV == 4 || NWV * PW4 <= NWV * 16 * LDH, "the partials of eight waves fit the A regions");
  if (NWV != 4) __syncthreads();
  {
    float* rw = red + wave * PW4;
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int c = 0; c < 2; ++c)
        st4(rw + (g * 128 + c * 64 + lane) * 4, (SG_LDS && g * 2 + c < 5) ? ld4(sgl + ((g * 2 + c) * 64 + lane) * 4) : f32x4{sg[g][c][0], sg[g][c][1], sg[g][c][2], sg[g][c][3]});
    const float* sm = Small + wave * SMALL_W;
    if (lane < 32) st4(rw + (384 + lane) * 4, f32x4{sm[132 + lane], sm[164 + lane], sm[196 + lane], sm[96 + lane]});
    for (int o = lane; o < 96; o += 64) rw[1664 + o] = sm[o];
    if (lane < 3) rw[1760 + lane] = sm[128 + lane];
    if (lane == 0) rw[1763] = 0.f;
  }
  __syncthreads();
  for (int o = tid; o < PW4; o += 64 * NWV) {
    float a = red[o] + red[PW4 + o] + red[2 * PW4 + o] + red[3 * PW4 + o];
    if (NWV == 8) a += red[4 * PW4 + o] + red[5 * PW4 + o] + red[6 * PW4 + o] + red[7 * PW4 + o];
    p.partial[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * PW4 + o] = a;
  }
#endif
}

// ---- the small gradients from the [416][4] sums ------------------------------------------------------------------------------
struct FinParams {
  const float* S;        // [PW4] (column sums of the workgroup partials)
  const float *w_off, *b_off, *w_zr, *w_q, *w_1;   // fp32 parameters
  float* dW_gates;       // [384][192]: columns 128..191 written here (0..127 by the weight-gradient pass)
  float* dW1;            // [32][192]: columns 128..191
  float* dW_off;         // [64][3]
  float* db_off;         // [64]
  float* db;             // [416]: d b_z | d b_r | d b_q | d b_1
};
__global__ __launch_bounds__(256) void gru_lean_finalize_kernel(FinParams p) {
  __shared__ float S[XT_ROWS * 4];
  const int tid = threadIdx.x;
  for (int i = tid; i < XT_ROWS * 4; i += 256) S[i] = p.S[i];
  __syncthreads();
  if (blockIdx.x == 0) {
    // offset encoder: dW_off[j][d] = sum_o W[o][128 + j] S[o][d], d b_off[j] = sum_o W[o][128 + j] S[o][3]
    const int j = tid >> 2, d = tid & 3;
    double a = 0.;
    for (int o = 0; o < XT_ROWS; ++o) {
      const float* wrow = o < 256 ? p.w_zr + o * 192 : o < 384 ? p.w_q + (o - 256) * 192 : p.w_1 + (o - 384) * 192;
      a += (double)wrow[128 + j] * (double)S[o * 4 + d];
    }
    if (d < 3) p.dW_off[j * 3 + d] = (float)a;
    else p.db_off[j] = (float)a;
    for (int o = tid; o < XT_ROWS; o += 256) p.db[o] = S[o * 4 + 3];
    return;
  }
  // x columns of the gate / head weight gradients: dW[o][128 + j] = S[o][:3] . W_off[j][:] + S[o][3] b_off[j]
  for (int idx = (blockIdx.x - 1) * 256 + tid; idx < XT_ROWS * 64; idx += (gridDim.x - 1) * 256) {
    const int o = idx >> 6, j = idx & 63;
    const float v = fmaf(S[o * 4 + 0], p.w_off[j * 3 + 0], fmaf(S[o * 4 + 1], p.w_off[j * 3 + 1],
                    fmaf(S[o * 4 + 2], p.w_off[j * 3 + 2], S[o * 4 + 3] * p.b_off[j])));
    if (o < 384) p.dW_gates[o * 192 + 128 + j] = v;
    else p.dW1[(o - 384) * 192 + 128 + j] = v;
  }
}

// Waves per workgroup of the lean kernels (round 6; same arithmetic per point in every form, bit-identical results):
//   backward: DF_GRU_WAVES = 8 (default: 128 points, one workgroup per CU, five of the six S accumulators in LDS: no scratch traffic
//             in the iteration loop -- 5.26 vs 5.60 ms) | 4 (64 points, two workgroups per CU: rounds 3-5; 20 scratch instructions per iteration)
//   forward:  DF_GRU_FWD_WAVES = 12 (default in bf16x2 mode: 192 points, THREE waves per SIMD at 168 registers -- 2.50 vs 2.72 ms,
//             tools/bench_gru_lean.py) | 8 | 4
int gru_waves() {
  const char* e = getenv("DF_GRU_WAVES");
  return (e && atoi(e) == 4) ? 4 : 8;
}

bool img64_ok4(const df_img& d, int B) {
  return d.ptr && df_aligned16(d.ptr) && d.n == B && d.c == 64 && (d.ld % 4) == 0 && (d.img_stride % 4) == 0 &&
         (d.grp_off % 4) == 0 && d.grp_size > 0;
}

}  // namespace

extern "C" int df_gru_xtab(df_gru_weights wts, float* xtab, void* stream) {
  DF_REQUIRE(wts.w_off && wts.b_off && wts.w_zr && wts.b_zr && wts.w_q && wts.b_q && wts.w_1 && wts.b_1 && xtab && df_aligned16(xtab),
             DF_E_ARG);
  XtabParams p{wts.w_off, wts.b_off, wts.w_zr, wts.b_zr, wts.w_q, wts.b_q, wts.w_1, wts.b_1, xtab};
  hipLaunchKernelGGL(gru_xtab_kernel, dim3((XT_ROWS + 63) / 64), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), p);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

extern "C" int df_gru_lean_partial_width(void) { return PW4; }

extern "C" int df_gru_lean_fwd(df_img before, df_img after, const int32_t* coords, const float* offs, const int32_t* counts, int B,
                               int N, int num_iters, df_gru_weights wts, const float* xtab, float* flow, float* hsave, int mfma_bf16,
                               void* stream) {
  DF_REQUIRE(img64_ok4(before, B) && img64_ok4(after, B), DF_E_SHAPE);
  DF_REQUIRE(before.h == after.h && before.w == after.w, DF_E_SHAPE);
  DF_REQUIRE(coords && offs && counts && flow && xtab && B > 0 && N > 0 && num_iters >= 1, DF_E_ARG);
  DF_REQUIRE(mfma_bf16 >= 0 && mfma_bf16 <= 3, DF_E_ARG);
  DF_REQUIRE(wts.w_zr && wts.w_q && wts.w_1 && wts.w_2 && wts.b_2 && df_aligned16(wts.w_zr) && df_aligned16(wts.w_q) &&
                 df_aligned16(wts.w_1) && df_aligned16(xtab) && (!hsave || df_aligned16(hsave)),
             DF_E_ARG);
  // (the weight-stationary forward experiment of round 5 is NOT part of the library: tools/archive/experiments/decoder5_ws.hip -- exact only
  //  with matrix-pipe drains after every tile, then no faster than this kernel, and without them one wrong tile in some launches)
  Gru4Params p;
  p.before = before; p.after = after; p.coords = coords; p.offs = offs; p.counts = counts;
  p.N = N; p.T = num_iters; p.w = wts; p.xtab = xtab; p.flow = flow; p.hsave = hsave;
  p.iter_stride = (int64_t)B * N * 128;
  static const int nwv0 = getenv("DF_GRU_FWD_WAVES") ? atoi(getenv("DF_GRU_FWD_WAVES")) : 12;
  const int nwv = (nwv0 == 12 && mfma_bf16 != 3) ? gru_waves() : (nwv0 == 12 || nwv0 == 8) ? nwv0 : 4;
  const dim3 grid((N + 16 * nwv - 1) / (16 * nwv), B);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define DF_FWD4(M)                                                                                      \
  do {                                                                                                  \
    if (nwv == 12 && M == 3) {                                                                          \
      if (hsave) hipLaunchKernelGGL((gru_fwd4_kernel<true, 3, 12>), grid, dim3(768), 0, s, p);          \
      else hipLaunchKernelGGL((gru_fwd4_kernel<false, 3, 12>), grid, dim3(768), 0, s, p);               \
    } else if (nwv == 8) {                                                                              \
      if (hsave) hipLaunchKernelGGL((gru_fwd4_kernel<true, M, 8>), grid, dim3(512), 0, s, p);           \
      else hipLaunchKernelGGL((gru_fwd4_kernel<false, M, 8>), grid, dim3(512), 0, s, p);                \
    } else {                                                                                            \
      if (hsave) hipLaunchKernelGGL((gru_fwd4_kernel<true, M, 4>), grid, dim3(256), 0, s, p);           \
      else hipLaunchKernelGGL((gru_fwd4_kernel<false, M, 4>), grid, dim3(256), 0, s, p);                \
    }                                                                                                   \
  } while (0)
  switch (mfma_bf16) {
    case 0: DF_FWD4(0); break;
    case 1: DF_FWD4(1); break;
    case 2: DF_FWD4(2); break;
    default: DF_FWD4(3); break;
  }
#undef DF_FWD4
  DF_CHECK_LAUNCH();
  return DF_OK;
}

extern "C" int df_gru_lean_bwd(const float* dflow, const float* offs, const int32_t* counts, int B, int N, int num_iters,
                               df_gru_weights wts, df_gru_weights_t wtt, const float* xtab, const float* hsave, float* gplanes,
                               float* dh0, float* dpre1, float* partial, int mfma_bf16, void* stream) {
  DF_REQUIRE(dflow && offs && counts && xtab && hsave && gplanes && dh0 && dpre1 && partial && B > 0 && N > 0 && num_iters >= 1,
             DF_E_ARG);
  DF_REQUIRE(mfma_bf16 >= 0 && mfma_bf16 <= 3, DF_E_ARG);
  DF_REQUIRE(wts.w_zr && wts.w_q && wts.w_1 && wts.w_2 && wtt.wt_zr && wtt.wt_q && wtt.wt_1, DF_E_ARG);
  DF_REQUIRE(df_aligned16(wts.w_zr) && df_aligned16(wts.w_q) && df_aligned16(wts.w_1) && df_aligned16(wtt.wt_zr) &&
                 df_aligned16(wtt.wt_q) && df_aligned16(wtt.wt_1) && df_aligned16(xtab) && df_aligned16(hsave) &&
                 df_aligned16(gplanes) && df_aligned16(dh0),
             DF_E_ALIGN);
  GruBwd4Params p;
  p.dflow = dflow; p.offs = offs; p.counts = counts; p.N = N; p.T = num_iters; p.w = wts; p.wt = wtt; p.xtab = xtab;
  p.hsave = hsave; p.gplanes = gplanes;
  p.iter_stride = (int64_t)B * N * 128;
  p.plane_stride = p.iter_stride * num_iters;
  p.dh0 = dh0; p.dpre1 = dpre1; p.partial = partial;
  static const int nwv = gru_waves();
  const dim3 grid((N + 16 * nwv - 1) / (16 * nwv), B);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define DF_BWD4(M)                                                                                      \
  do {                                                                                                  \
    if (nwv == 8) hipLaunchKernelGGL((gru_bwd4_kernel<M, 8>), grid, dim3(512), 0, s, p);                \
    else hipLaunchKernelGGL((gru_bwd4_kernel<M, 4>), grid, dim3(256), 0, s, p);                         \
  } while (0)
  switch (mfma_bf16) {
    case 0: DF_BWD4(0); break;
    case 1: DF_BWD4(1); break;
    case 2: DF_BWD4(2); break;
    default: DF_BWD4(3); break;
  }
#undef DF_BWD4
  DF_CHECK_LAUNCH();
  return DF_OK;
}

extern "C" int df_gru_lean_finalize(const float* sums, df_gru_weights wts, float* dW_gates, float* dW1, float* dW_off, float* db_off,
                                    float* db, void* stream) {
  DF_REQUIRE(sums && wts.w_off && wts.b_off && wts.w_zr && wts.w_q && wts.w_1 && dW_gates && dW1 && dW_off && db_off && db, DF_E_ARG);
  FinParams p{sums, wts.w_off, wts.b_off, wts.w_zr, wts.w_q, wts.w_1, dW_gates, dW1, dW_off, db_off, db};
  hipLaunchKernelGGL(gru_lean_finalize_kernel, dim3(1 + 26), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
  DF_CHECK_LAUNCH();
  return DF_OK;
}
