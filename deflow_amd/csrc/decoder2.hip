// Second-generation ConvGRU decoder forward ([REF decoder.py:123-183]) built for TWO workgroups per CU.
//
// The first-generation kernel (decoder.hip) keeps 424 registers and 124 KB of LDS per workgroup: one wave per SIMD,
// so gate math, saved-plane stores and barriers are never overlapped with another wave's MFMAs (measured 60 % of the
// fp32 MFMA peak in inference, 48 % with the training saves).  Changes here, same arithmetic:
//   * the [z | r] GEMM is two 128-row GEMMs: 32 accumulator registers live instead of 64 and a 128-row weight tile
//     (2 x 18 KB of LDS instead of 2 x 36 KB);
//   * the loop-invariant x operand (offset encoding, 64 of the 192 k columns) lives in REGISTERS as four MFMA A
//     fragments per lane, loaded once -- the wave-private LDS A region holds only h / r*h (16 x 128);
//   * weight chunks are register-staged with 4 float4 per thread.
// => 70.7 KB LDS and <= 256 VGPRs per workgroup: two workgroups per CU (2 waves / SIMD).
#include "common.h"
#include "gemm_stream.h"

namespace {

using namespace gs;

constexpr int LDH = 132;              // A region pitch: 33 slots of 16 B, 33 mod 16 = 1 -> conflict-free b128 rows
constexpr int BS2 = 128 * LDB;        // one B buffer: 128 weight rows x (32 + 4) floats

// Weight chunk stager with a 32-bit per-thread offset: the (uniform) matrix base stays in SGPRs and each load is
// base + voffset, instead of one 64-bit vector address per row group (the latter cost ~60 spilled registers).
struct Stager2 {
  f32x4 r[4];
};
// Buffer addressing (SGPR resource + one 32-bit VGPR offset + SGPR offset) for everything the decoder kernels stream:
// no 64-bit vector addresses, and the resource's byte range is the bounds check -- loads past it return 0, stores are
// dropped -- so partially filled 16-row wave tiles need no per-row predicates.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_ld4(rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ float buf_ld1(rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_st4(rsrc_t r, unsigned voff, unsigned soff, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, 0);
}
__device__ __forceinline__ void buf_st1(rsrc_t r, unsigned voff, unsigned soff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
}

// 32-deep k chunk `chunk` of ROWS weight rows (LDW floats apart) -> 4 float4 per thread
template <int ROWS, int LDW>
__device__ __forceinline__ void stage_load_t(Stager2& s, const float* __restrict__ W, int chunk) {
  const rsrc_t r = make_rsrc(W, 0x7fffffffu);
  const unsigned voff = ((threadIdx.x >> 3) * (unsigned)LDW + (threadIdx.x & 7) * 4u) * 4u;
#pragma unroll
  for (int i = 0; i < ROWS / 32; ++i) s.r[i] = buf_ld4(r, voff, (i * 32 * LDW + chunk * 32) * 4);
}
template <int ROWS>
__device__ __forceinline__ void stage_load2(Stager2& s, const float* __restrict__ W, int chunk) {
  stage_load_t<ROWS, 192>(s, W, chunk);
}
template <int ROWS>
__device__ __forceinline__ void stage_store2(const Stager2& s, float* Bbuf) {
  const unsigned off = (threadIdx.x >> 3) * LDB + (threadIdx.x & 7) * 4u;
#pragma unroll
  for (int i = 0; i < ROWS / 32; ++i) st4(Bbuf + i * 32 * LDB + off, s.r[i]);
}

// acc[t] += A[16, 192] * W[rows, 192]^T for a 16-row wave tile; k columns 0..127 come from the LDS A region, 128..191
// from the register fragments xf.  Chunk 0 of W must already be in buffer `par`; the first chunk of Wnext is
// prefetched during the last chunk (same contract as gs::gemm_stream).
template <int ROWS, int ROWS_NEXT>
__device__ __forceinline__ void gemm_hx(const float* __restrict__ W, const float* __restrict__ Wnext, const float* a_lane,
                                        const f32x4 (&xf)[4], float* Bs, int& par, f32x4 (&acc)[ROWS / 16], Stager2& stg) {
  const int lane = threadIdx.x & 63;
  const float* b_lane = Bs + (lane & 15) * LDB + (lane >> 4) * 4;
  constexpr int NPAIR = ROWS / 32;
  auto chunk = [&](int c, const f32x4 a0, const f32x4 a1) {
    if (c + 1 < 6) stage_load2<ROWS>(stg, W, c + 1);
    else if (Wnext) stage_load2<ROWS_NEXT>(stg, Wnext, 0);
    const float* bb = b_lane + ((par + c) & 1) * BS2;
    f32x4 nb0 = ld4(bb), nb1 = ld4(bb + 16 * LDB);
#pragma unroll
    for (int j = 0; j < 2 * NPAIR; ++j) {
      const int g = j / NPAIR, t = 2 * (j % NPAIR);
      const f32x4 b0 = nb0, b1 = nb1;
      if (j + 1 < 2 * NPAIR) {
        const int gn = (j + 1) / NPAIR, tn = 2 * ((j + 1) % NPAIR);
        nb0 = ld4(bb + tn * 16 * LDB + gn * 16);
        nb1 = ld4(bb + (tn + 1) * 16 * LDB + gn * 16);
      }
      const f32x4 a = g ? a1 : a0;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b0[s], acc[t], 0, 0, 0);
        acc[t + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b1[s], acc[t + 1], 0, 0, 0);
      }
    }
    float* nb = Bs + ((par + c + 1) & 1) * BS2;
    if (c + 1 < 6) stage_store2<ROWS>(stg, nb);
    else if (Wnext) stage_store2<ROWS_NEXT>(stg, nb);
    __syncthreads();
  };
#pragma unroll 1
  for (int c = 0; c < 4; ++c) chunk(c, ld4(a_lane + c * 32), ld4(a_lane + c * 32 + 16));
  chunk(4, xf[0], xf[1]);
  chunk(5, xf[2], xf[3]);
  // six chunks: the buffer parity is unchanged
}

struct Gru2Params {
  df_img before, after;
  const int32_t* coords;
  const float* offs;
  const int32_t* counts;
  int N, T;
  df_gru_weights w;
  float* flow;
  float* save;
  int64_t plane_stride, iter_stride;
};

template <bool SAVE>
__global__ __launch_bounds__(256, 2) void gru_fwd2_kernel(Gru2Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(16))) float Bs[2 * BS2];       // 36.9 KB
  __shared__ __attribute__((aligned(16))) float As[4 * 16 * LDH];  // 33.8 KB
  const int b = blockIdx.y;
  const int cnt = p.counts[b];
  const int p0 = blockIdx.x * 64;
  if (p0 >= cnt) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform -> SGPR addressing below
  const int li = lane & 15, lq = lane >> 4;
  float* Aw = As + wave * 16 * LDH;
  const int wp0 = p0 + wave * 16;
  const int64_t grow0 = (int64_t)b * p.N + wp0;
  const float* a_lane = Aw + li * LDH + lq * 4;
  float* c_lane = Aw + 4 * lq * LDH + li;            // C-layout element (row 4 lq + r, col 16 t + li) = c_lane[r * LDH + 16 t]
  float* r_lane = Aw + (lane >> 5) * LDH + (lane & 31) * 4;  // row copies: float4 j at r_lane + 2 j LDH
  const unsigned row_bytes = (unsigned)min(max(cnt - wp0, 0), 16) * 512u;  // valid part of this wave's [16][128] tile
  const unsigned rl_off = ((lane >> 5) * 128 + (lane & 31) * 4) * 4, cl_off = (4 * lq * 128 + li) * 4;
  const float* w_z = p.w.w_zr;
  const float* w_r = p.w.w_zr + 128 * 192;

  Stager2 stg;
  int par = 0;
  stage_load2<128>(stg, w_z, 0);

  // ---- x = offset encoder -> A region (temporarily) -> register fragments -----------------------------------
  f32x4 xf[4];
  {
    const float w0 = p.w.w_off[lane * 3 + 0], w1 = p.w.w_off[lane * 3 + 1], w2 = p.w.w_off[lane * 3 + 2];
    const float bo = p.w.b_off[lane];
    for (int pt = 0; pt < 16; ++pt) {
      float x = 0.f;
      if (wp0 + pt < cnt) {
        const float* o = p.offs + (grow0 + pt) * 3;
        x = fmaf(w2, o[2], fmaf(w1, o[1], fmaf(w0, o[0], bo)));
      }
      Aw[pt * LDH + lane] = x;
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) xf[k] = ld4(a_lane + k * 16);   // k columns 128 + 16k + 4 lq .. +3 of row li
  __syncthreads();
  // ---- gather h0 = [before | after] ------------------------------------------------------------------------------
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
  stage_store2<128>(stg, Bs);
  __syncthreads();

  f32x4 h[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) h[t][r] = c_lane[r * LDH + 16 * t];

  auto save_rows = [&](int plane, int it) {  // coalesced copy of the wave's 16 x 128 A region
    const rsrc_t dst = make_rsrc(p.save + plane * p.plane_stride + it * p.iter_stride + grow0 * 128, row_bytes);
#pragma unroll
    for (int j = 0; j < 8; ++j) buf_st4(dst, rl_off + j * 1024, 0, ld4(r_lane + 2 * j * LDH));
  };
  auto save_regs = [&](int plane, int it, const f32x4 (&v)[8]) {  // C-layout registers -> [row][128]
    const rsrc_t dst = make_rsrc(p.save + plane * p.plane_stride + it * p.iter_stride + grow0 * 128, row_bytes);
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) buf_st1(dst, cl_off + (r * 128 + 16 * t) * 4, 0, v[t][r]);
  };
  auto bias_init = [&](f32x4 (&acc)[8], const float* bias) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float bia = bias[16 * t + li];
      acc[t] = f32x4{bia, bia, bia, bia};
    }
  };

  for (int it = 0; it < p.T; ++it) {
    if (SAVE) save_rows(0, it);  // h_in
    f32x4 z[8], acc[8];
    bias_init(z, p.w.b_zr);
    gemm_hx<128, 128>(w_z, w_r, a_lane, xf, Bs, par, z, stg);
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) z[t][r] = df_sigmoid_fast(z[t][r]);
    if (SAVE) save_regs(1, it, z);
    bias_init(acc, p.w.b_zr + 128);
    gemm_hx<128, 128>(w_r, p.w.w_q, a_lane, xf, Bs, par, acc, stg);
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t][r] = df_sigmoid_fast(acc[t][r]);
    if (SAVE) save_regs(2, it, acc);
    // every wave has finished reading h from its A region (barrier at the end of the GEMM): overwrite with r * h
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) c_lane[r * LDH + 16 * t] = acc[t][r] * h[t][r];
    __syncthreads();
    if (SAVE) save_rows(4, it);  // r * h
    bias_init(acc, p.w.b_q);
    const bool last = it + 1 == p.T;
    if (!last) gemm_hx<128, 128>(p.w.w_q, w_z, a_lane, xf, Bs, par, acc, stg);
    else gemm_hx<128, 32>(p.w.w_q, p.w.w_1, a_lane, xf, Bs, par, acc, stg);
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc[t][r] = df_tanh_fast(acc[t][r]);
        h[t][r] = (1.f - z[t][r]) * h[t][r] + z[t][r] * acc[t][r];
        c_lane[r * LDH + 16 * t] = h[t][r];
      }
    if (SAVE) save_regs(3, it, acc);
    __syncthreads();
  }
  if (SAVE) save_rows(5, 0);  // h_T
  // ---- MLP head ----------------------------------------------------------------------------------------------------
  f32x4 hid[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float bia = p.w.b_1[16 * t + li];
    hid[t] = f32x4{bia, bia, bia, bia};
  }
  gemm_hx<32, 32>(p.w.w_1, nullptr, a_lane, xf, Bs, par, hid, stg);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) c_lane[r * LDH + 16 * t] = df_gelu(hid[t][r]);
  __syncthreads();
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

}  // namespace

// Arguments are validated by the C-ABI entry (df_gru_decoder_fwd in decoder.hip), which dispatches here.
int df_launch_gru_fwd2(df_img before, df_img after, const int32_t* coords, const float* offs, const int32_t* counts,
                       int B, int N, int num_iters, df_gru_weights wts, float* flow, float* save, void* stream) {
  Gru2Params p;
  p.before = before; p.after = after; p.coords = coords; p.offs = offs; p.counts = counts;
  p.N = N; p.T = num_iters; p.w = wts; p.flow = flow; p.save = save;
  p.iter_stride = (int64_t)B * N * 128;
  p.plane_stride = p.iter_stride * num_iters;
  const dim3 grid((N + 63) / 64, B);
  if (save) hipLaunchKernelGGL(gru_fwd2_kernel<true>, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
  else hipLaunchKernelGGL(gru_fwd2_kernel<false>, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

// =====================================================================================================================
// Backward, same design ([REF decoder.py:123-183] differentiated; first generation: decoder_bwd.hip).
//
//   * dh is accumulated IN PLACE: dh <- dh (1 - z), then the transposed-weight GEMMs add W_z^T dz_pre, r * (W_q^T dq_pre)
//     and W_r^T dr_pre into the same registers -- at most four 32-register planes are live at any point;
//   * GEMM order per step: dz_pre (needs only h, z, q), dq_pre, then dr_pre -- the A region (16 x 128) holds one gate
//     gradient at a time and each is copied to its saved plane (coalesced rows) while it sits there;
//   * every [192 -> 128 | 64] output is two GEMMs over the same A operand (128 weight rows, then 64): 128-row B tiles;
//   * the A region is wave-private, so its write -> read hand-offs need only wave-level ordering (LDS executes one
//     wave's instructions in order); workgroup barriers remain only around the shared weight buffers.
namespace {

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// acc[t] += A[16, 32 NCH] * W[ROWS, 32 NCH]^T with W rows LDW floats apart (a window of a transposed weight matrix).
template <int ROWS, int LDW, int NCH, int ROWS_NEXT, int LDW_NEXT>
__device__ __forceinline__ void gemm_t(const float* __restrict__ W, const float* __restrict__ Wnext, const float* a_lane,
                                       float* Bs, int& par, f32x4 (&acc)[ROWS / 16], Stager2& stg) {
  const int lane = threadIdx.x & 63;
  const float* b_lane = Bs + (lane & 15) * LDB + (lane >> 4) * 4;
  constexpr int NPAIR = ROWS / 32;
#pragma unroll 1
  for (int c = 0; c < NCH; ++c) {
    const bool more = c + 1 < NCH;
    if (more) stage_load_t<ROWS, LDW>(stg, W, c + 1);
    else if (Wnext) stage_load_t<ROWS_NEXT, LDW_NEXT>(stg, Wnext, 0);
    const float* bb = b_lane + ((par + c) & 1) * BS2;
    const f32x4 a0 = ld4(a_lane + c * 32), a1 = ld4(a_lane + c * 32 + 16);
    f32x4 nb0 = ld4(bb), nb1 = ld4(bb + 16 * LDB);
#pragma unroll
    for (int j = 0; j < 2 * NPAIR; ++j) {
      const int g = j / NPAIR, t = 2 * (j % NPAIR);
      const f32x4 b0 = nb0, b1 = nb1;
      if (j + 1 < 2 * NPAIR) {
        const int gn = (j + 1) / NPAIR, tn = 2 * ((j + 1) % NPAIR);
        nb0 = ld4(bb + tn * 16 * LDB + gn * 16);
        nb1 = ld4(bb + (tn + 1) * 16 * LDB + gn * 16);
      }
      const f32x4 a = g ? a1 : a0;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b0[s], acc[t], 0, 0, 0);
        acc[t + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b1[s], acc[t + 1], 0, 0, 0);
      }
    }
    float* nb = Bs + ((par + c + 1) & 1) * BS2;
    if (more) stage_store2<ROWS>(stg, nb);
    else if (Wnext) stage_store2<ROWS_NEXT>(stg, nb);
    __syncthreads();
  }
  par = (par + NCH) & 1;
}

struct GruBwd2Params {
  const float* dflow;
  const float* offs;
  const int32_t* counts;
  int N, T;
  df_gru_weights w;
  df_gru_weights_t wt;
  float* save;
  int64_t plane_stride, iter_stride;
  float* dh0;
  float* dx;
  float* dpre1;
  float* xout;
  float* bias_partial;  // [blocks][772], layout as in decoder_bwd.hip
};

__global__ __launch_bounds__(256, 2) void gru_bwd2_kernel(GruBwd2Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(16))) float Bs[2 * BS2];
  __shared__ __attribute__((aligned(16))) float As[4 * 16 * LDH];
  constexpr int SMALL_W = 132;
  __shared__ float Small[4 * SMALL_W];
  const int b = blockIdx.y;
  const int cnt = p.counts[b];
  const int p0 = blockIdx.x * 64;
  if (p0 >= cnt) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  float* Aw = As + wave * 16 * LDH;
  const int wp0 = p0 + wave * 16;
  const int64_t grow0 = (int64_t)b * p.N + wp0;
  const float* a_lane = Aw + li * LDH + lq * 4;
  float* c_lane = Aw + 4 * lq * LDH + li;            // C-layout element (row 4 lq + r, col 16 t + li) = c_lane[r * LDH + 16 t]
  float* r_lane = Aw + (lane >> 5) * LDH + (lane & 31) * 4;  // row copies: float4 j at r_lane + 2 j LDH
  const unsigned nvalid = (unsigned)min(max(cnt - wp0, 0), 16);  // valid rows of this wave's 16-row tile
  const unsigned row_bytes = nvalid * 512u;
  const unsigned rl_off = ((lane >> 5) * 128 + (lane & 31) * 4) * 4, cl_off = (4 * lq * 128 + li) * 4;
  const float* wt_q = p.wt.wt_q;
  const float* wt_zr = p.wt.wt_zr;

  Stager2 stg;
  int par = 0;
  stage_load2<32>(stg, p.w.w_1, 0);

  auto lds_to_rows = [&](float* dst) {  // the wave's 16 x 128 A region -> global rows (coalesced; invalid rows dropped)
    const rsrc_t d = make_rsrc(dst + grow0 * 128, row_bytes);
#pragma unroll
    for (int j = 0; j < 8; ++j) buf_st4(d, rl_off + j * 1024, 0, ld4(r_lane + 2 * j * LDH));
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
  // saved plane -> C-layout registers straight from global memory (64-byte segments per row group); rows beyond cnt
  // are outside the buffer range and read as 0
  auto c_load = [&](f32x4 (&v)[8], const float* src) {
    const rsrc_t s0 = make_rsrc(src + grow0 * 128, row_bytes);
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[t][r] = buf_ld1(s0, cl_off + (r * 128 + 16 * t) * 4, 0);
  };
  // ---- x (offset encoder): global copy for the weight-gradient GEMMs + register fragments for the first GEMM -------
  f32x4 xf[4];
  {
    const float w0 = p.w.w_off[lane * 3 + 0], w1 = p.w.w_off[lane * 3 + 1], w2 = p.w.w_off[lane * 3 + 2];
    const float bo = p.w.b_off[lane];
    for (int pt = 0; pt < 16; ++pt) {
      float x = 0.f;
      if (wp0 + pt < cnt) {
        const float* o = p.offs + (grow0 + pt) * 3;
        x = fmaf(w2, o[2], fmaf(w1, o[1], fmaf(w0, o[0], bo)));
        p.xout[(grow0 + pt) * 64 + lane] = x;
      }
      Aw[pt * LDH + lane] = x;
    }
    wave_lds_sync();
#pragma unroll
    for (int k = 0; k < 4; ++k) xf[k] = ld4(a_lane + k * 16);
    wave_lds_sync();
  }
  {  // h_T rows -> A region
    const rsrc_t s0 = make_rsrc(p.save + 5 * p.plane_stride + grow0 * 128, row_bytes);
#pragma unroll
    for (int j = 0; j < 8; ++j) st4(r_lane + 2 * j * LDH, buf_ld4(s0, rl_off + j * 1024, 0));
  }
  stage_store2<32>(stg, Bs);
  __syncthreads();

  // bias gradients of the three gates: column sums of the gate-gradient planes, taken from the A region while each
  // plane sits there (lane j owns columns j and j + 64): 6 accumulator registers instead of 24 per-lane partials
  float sb[3][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
  auto colsum = [&](int g) {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s0 += Aw[r * LDH + lane];
      s1 += Aw[r * LDH + 64 + lane];
    }
    sb[g][0] += s0;
    sb[g][1] += s1;
  };
  float sb1[2] = {0.f, 0.f};
  float sw2[2][3], sdf[3];
  // ---- MLP head backward -------------------------------------------------------------------------------------------
  f32x4 pre1[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float bia = p.w.b_1[16 * t + li];
    pre1[t] = f32x4{bia, bia, bia, bia};
  }
  {  // pre1 = [h_T | x] W1^T + b1; prefetches the first tile of W1^T (rows 0..127, 32 wide)
    const int lane_ = threadIdx.x & 63;
    const float* b_lane = Bs + (lane_ & 15) * LDB + (lane_ >> 4) * 4;
    auto chunk = [&](int c, const f32x4 a0, const f32x4 a1) {
      if (c + 1 < 6) stage_load2<32>(stg, p.w.w_1, c + 1);
      else stage_load_t<128, 32>(stg, p.wt.wt_1, 0);
      const float* bb = b_lane + ((par + c) & 1) * BS2;
      const f32x4 b00 = ld4(bb), b01 = ld4(bb + 16 * LDB), b10 = ld4(bb + 16), b11 = ld4(bb + 16 * LDB + 16);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        pre1[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], b00[s], pre1[0], 0, 0, 0);
        pre1[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], b01[s], pre1[1], 0, 0, 0);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        pre1[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], b10[s], pre1[0], 0, 0, 0);
        pre1[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], b11[s], pre1[1], 0, 0, 0);
      }
      float* nb = Bs + ((par + c + 1) & 1) * BS2;
      if (c + 1 < 6) stage_store2<32>(stg, nb);
      else stage_store2<128>(stg, nb);
      __syncthreads();
    };
#pragma unroll 1
    for (int c = 0; c < 4; ++c) chunk(c, ld4(a_lane + c * 32), ld4(a_lane + c * 32 + 16));
    chunk(4, xf[0], xf[1]);
    chunk(5, xf[2], xf[3]);
  }
  {
    float df[4][3];
    const rsrc_t dfl = make_rsrc(p.dflow + grow0 * 3, nvalid * 12u);
    const rsrc_t dp1 = make_rsrc(p.dpre1 + grow0 * 32, nvalid * 128u);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int o = 0; o < 3; ++o) df[r][o] = buf_ld1(dfl, (4 * lq + r) * 12 + o * 4, 0);
#pragma unroll
    for (int o = 0; o < 3; ++o) sdf[o] = df[0][o] + df[1][o] + df[2][o] + df[3][o];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int col = 16 * t + li;
      const float w20 = p.w.w_2[0 * 32 + col], w21 = p.w.w_2[1 * 32 + col], w22 = p.w.w_2[2 * 32 + col];
      sw2[t][0] = sw2[t][1] = sw2[t][2] = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pre = pre1[t][r];
        const float dhid = df[r][0] * w20 + df[r][1] * w21 + df[r][2] * w22;
        const float dp = dhid * df_gelu_grad(pre);
        const float hv = df_gelu(pre);
        buf_st1(dp1, ((4 * lq + r) * 32 + col) * 4, 0, dp);
#pragma unroll
        for (int o = 0; o < 3; ++o) sw2[t][o] += df[r][o] * hv;  // df is 0 on invalid rows
        c_lane[r * LDH + 16 * t] = dp;                       // A operand of the next GEMM (k = 0..31)
        if (r == 0) sb1[t] = dp; else sb1[t] += dp;
      }
    }
  }
  {  // the head's small sums leave the registers now: reduce over the 4 row groups and park them in LDS
    float* sm = Small + wave * SMALL_W;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int o = 0; o < 3; ++o) {
        float v = sw2[t][o];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (lq == 0) sm[o * 32 + 16 * t + li] = v;
      }
      float v = sb1[t];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (lq == 0) sm[96 + 16 * t + li] = v;
    }
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      float v = sdf[o];  // identical on the 16 lanes of a row group: reduce over the 4 row groups only
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (lane == 0) sm[128 + o] = v;
    }
  }
  wave_lds_sync();
  f32x4 dh[8], dxa[4];
#pragma unroll
  for (int t = 0; t < 8; ++t) dh[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 4; ++t) dxa[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  gemm_t<128, 32, 1, 64, 32>(p.wt.wt_1, p.wt.wt_1 + 128 * 32, a_lane, Bs, par, dh, stg);
  gemm_t<64, 32, 1, 128, 256>(p.wt.wt_1 + 128 * 32, wt_zr, a_lane, Bs, par, dxa, stg);

  // ---- GRU steps in reverse ----------------------------------------------------------------------------------------
  for (int it = p.T - 1; it >= 0; --it) {
    float* pl_h = p.save + 0 * p.plane_stride + it * p.iter_stride;
    float* pl_z = p.save + 1 * p.plane_stride + it * p.iter_stride;
    float* pl_r = p.save + 2 * p.plane_stride + it * p.iter_stride;
    float* pl_q = p.save + 3 * p.plane_stride + it * p.iter_stride;
    f32x4 h[8], q[8];
    {
      f32x4 z[8];
      c_load(h, pl_h);
      c_load(z, pl_z);
      c_load(q, pl_q);
      // h' = (1 - z) h + z q
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float d = dh[t][k], zz = z[t][k], qq = q[t][k];
          const float dzp = d * (qq - h[t][k]) * zz * (1.f - zz);
          dh[t][k] = d * (1.f - zz);
          q[t][k] = d * zz * (1.f - qq * qq);  // dq_pre
          z[t][k] = dzp;
        }
      c_to_lds(z);
    }
    wave_lds_sync();
    lds_to_rows(pl_z);  // dz_pre replaces z
    colsum(0);
    gemm_t<128, 256, 4, 64, 256>(wt_zr, wt_zr + 128 * 256, a_lane, Bs, par, dh, stg);
    gemm_t<64, 256, 4, 128, 128>(wt_zr + 128 * 256, wt_q, a_lane, Bs, par, dxa, stg);
    c_to_lds(q);
    wave_lds_sync();
    lds_to_rows(pl_q);  // dq_pre replaces q
    colsum(2);
    {
      f32x4 drh[8];
      c_load(q, pl_r);  // q <- r (lands during the GEMMs)
#pragma unroll
      for (int t = 0; t < 8; ++t) drh[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      gemm_t<128, 128, 4, 64, 128>(wt_q, wt_q + 128 * 128, a_lane, Bs, par, drh, stg);
      gemm_t<64, 128, 4, 128, 256>(wt_q + 128 * 128, wt_zr + 128, a_lane, Bs, par, dxa, stg);
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float d = drh[t][k], rr = q[t][k];
          dh[t][k] += d * rr;
          q[t][k] = d * h[t][k] * rr * (1.f - rr);  // dr_pre
        }
    }
    c_to_lds(q);
    wave_lds_sync();
    lds_to_rows(pl_r);  // dr_pre replaces r
    colsum(1);
    gemm_t<128, 256, 4, 64, 256>(wt_zr + 128, wt_zr + 128 * 256 + 128, a_lane, Bs, par, dh, stg);
    if (it > 0) gemm_t<64, 256, 4, 128, 256>(wt_zr + 128 * 256 + 128, wt_zr, a_lane, Bs, par, dxa, stg);
    else gemm_t<64, 256, 4, 128, 256>(wt_zr + 128 * 256 + 128, nullptr, a_lane, Bs, par, dxa, stg);
  }
  // ---- outputs: dh0 [rows,128], dx [rows,64] ----------------------------------------------------------------------
  c_to_lds(dh);
  wave_lds_sync();
  lds_to_rows(p.dh0);
  wave_lds_sync();
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int k = 0; k < 4; ++k) c_lane[k * LDH + 16 * t] = dxa[t][k];
  wave_lds_sync();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int f = lane + 64 * j;
    const int pt = f >> 4, c4 = f & 15;
    if (wp0 + pt < cnt) st4(p.dx + (grow0 + pt) * 64 + c4 * 4, ld4(r_lane + 2 * j * LDH));
  }
  // ---- per-workgroup partial sums of every small gradient (rows beyond cnt contributed exact zeros) ----------------
  constexpr int PW = 772;
  float* red = Bs;  // the weight buffers are idle now: [4 waves][PW]
  {
    float off[4][3];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int d = 0; d < 3; ++d) off[k][d] = (wp0 + 4 * lq + k < cnt) ? p.offs[(grow0 + 4 * lq + k) * 3 + d] : 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int d = 0; d < 3; ++d) v[d] += dxa[t][k] * off[k][d];
        v[3] += dxa[t][k];
      }
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        v[d] += __shfl_xor(v[d], 16);
        v[d] += __shfl_xor(v[d], 32);
      }
      if (lq == 0) {
        const int c = 16 * t + li;
        red[wave * PW + 416 + c * 3 + 0] = v[0];
        red[wave * PW + 416 + c * 3 + 1] = v[1];
        red[wave * PW + 416 + c * 3 + 2] = v[2];
        red[wave * PW + 608 + c] = v[3];
      }
    }
    const float* sm = Small + wave * SMALL_W;
    for (int o = lane; o < 96; o += 64) red[wave * PW + 672 + o] = sm[o];
    if (lane < 32) red[wave * PW + 384 + lane] = sm[96 + lane];
    if (lane < 3) red[wave * PW + 768 + lane] = sm[128 + lane];
    if (lane == 0) red[wave * PW + 771] = 0.f;
  }
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    red[wave * PW + g * 128 + lane] = sb[g][0];
    red[wave * PW + g * 128 + 64 + lane] = sb[g][1];
  }
  __syncthreads();
  for (int o = tid; o < PW; o += 256)
    p.bias_partial[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * PW + o] =
        red[o] + red[PW + o] + red[2 * PW + o] + red[3 * PW + o];
#endif
}

}  // namespace

// Arguments are validated by the C-ABI entry (df_gru_decoder_bwd in decoder_bwd.hip), which dispatches here.
int df_launch_gru_bwd2(const float* dflow, const float* offs, const int32_t* counts, int B, int N, int num_iters,
                       df_gru_weights wts, df_gru_weights_t wtt, float* save, float* dh0, float* dx, float* dpre1,
                       float* xout, float* bias_partial, void* stream) {
  GruBwd2Params p;
  p.dflow = dflow; p.offs = offs; p.counts = counts; p.N = N; p.T = num_iters; p.w = wts; p.wt = wtt; p.save = save;
  p.iter_stride = (int64_t)B * N * 128;
  p.plane_stride = p.iter_stride * num_iters;
  p.dh0 = dh0; p.dx = dx; p.dpre1 = dpre1; p.xout = xout; p.bias_partial = bias_partial;
  hipLaunchKernelGGL(gru_bwd2_kernel, dim3((N + 63) / 64, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
  DF_CHECK_LAUNCH();
  return DF_OK;
}
