// ConvGRU decoder forward with the loop-invariant x projection hoisted ([REF decoder.py:123-183]).
//
// x (the offset encoding) is the same in every GRU iteration [REF decoder.py:180], so W[:, 128:] x + b of the three
// gates is computed ONCE per point (registers, 3 x 32 per lane in MFMA C layout) and each iteration only multiplies
// the 128 h / r*h columns: 54 instead of 72 weight chunks for 4 iterations (-25 % MFMA work; only the fp32 summation
// order changes -- x part first, then the h part).  Layout: two workgroups per CU (70 KB LDS, <= 256 VGPRs each),
// 16 points per wave, wave-private LDS A operand), except that weight chunks now go L2 -> LDS by DMA
// (buffer_load ... lds): no staging registers, which is what makes room for the 96 hoisted ones.  The DMA writes
// each wave's 64 x 16 B linearly, so weight tiles are unpadded [rows][32] with the XOR slot swizzle of the conv
// kernels applied to the SOURCE column (physical slot = logical slot ^ ((row >> 1) & 7)).
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

struct Gru3Params {
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

// W16 (with BF): the gate / head weights p.w.{w_zr, w_q, w_1} point at bf16 copies (gemm_dma.h, WStreamT<2>)
// X2 (fp32 training, mfma_bf16 == 3): the same pointers hold the pre-split two-plane weights of WStreamT<3>; planes saved fp32
template <bool SAVE, bool BF = false, bool W16 = false, bool X2 = false>
__global__ __launch_bounds__(256, DF_GRU_LB) DF_GRU_ATTR void gru_fwd3_kernel(Gru3Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(16))) float Bs[2 * BT];         // 32 KB
  __shared__ __attribute__((aligned(16))) float As[4 * 16 * LDH];   // 33.8 KB
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
  static_assert(!X2 || (!BF && !W16), "X2 is a mode of its own");
  constexpr bool K8 = W16 || X2;                            // a lane's two k groups are adjacent (k = 8 lq .. 8 lq + 7)
  constexpr int WSC = W16 ? 2 : 1;                          // weight ROW offsets in floats: halved for bf16 data (X2: the fp32 pitch)
  const float* a_lane = Aw + li * LDH + lq * (K8 ? 8 : 4);
  float* c_lane = Aw + 4 * lq * LDH + li;                   // C-layout element (row 4 lq + r, col 16 t + li)
  float* r_lane = Aw + (lane >> 5) * LDH + (lane & 31) * 4;  // row copies: float4 j at r_lane + 2 j LDH
  const unsigned row_bytes = (unsigned)min(max(cnt - wp0, 0), 16) * 512u;
  const unsigned rl_off = ((lane >> 5) * 128 + (lane & 31) * 4) * 4, cl_off = (4 * lq * 128 + li) * 4;
  const float* w_z = p.w.w_zr;
  const float* w_r = p.w.w_zr + 128 * 192 / WSC;
  const float* w_q = p.w.w_q;

  WStreamT<X2 ? 3 : BF ? (W16 ? 2 : 1) : 0> ws;
  wstream_init(ws, Bs);
  dma_first<128, 192>(w_z, 4, Bs, ws);   // first chunk of the x projection

  // ---- x = offset encoder -> A region (temporarily) -> register fragments ------------------------------------------
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
  for (int k = 0; k < 4; ++k) xf[k] = ld4(a_lane + (K8 ? (k >> 1) * 32 + (k & 1) * 4 : k * 16));
  __syncthreads();
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

  auto bias_init = [&](f32x4 (&acc)[8], const float* bias) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float bia = bias[16 * t + li];
      acc[t] = f32x4{bia, bia, bia, bia};
    }
  };
  // ---- hoisted x projections (bias included) ---------------------------------------------------------------------
  f32x4 xz[8], xr[8], xq[8];
  bias_init(xz, p.w.b_zr);
  gemm<128, 2, true, 128>(w_z, 4, w_r, 4, a_lane, xf, ws, xz);
  bias_init(xr, p.w.b_zr + 128);
  gemm<128, 2, true, 128>(w_r, 4, w_q, 4, a_lane, xf, ws, xr);
  bias_init(xq, p.w.b_q);
  gemm<128, 2, true, 128>(w_q, 4, w_z, 0, a_lane, xf, ws, xq);

  f32x4 h[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) h[t][r] = c_lane[r * LDH + 16 * t];

  // Saved planes.  bf16 mode: planes 0..4 (h_in, z, r, q, r*h -- read back by the backward kernels, which round them to bf16
  // operands or use them in elementwise gradient formulas) are stored as bf16 in the first half of each row's 512-byte slot:
  // half the bytes of the kernels' dominant HBM stream; plane 5 (h_T, also the fp32 operand of the head's generic weight
  // gradient) stays fp32.
  auto save_rows = [&](int plane, int it) {  // coalesced copy of the wave's 16 x 128 A region
    const rsrc_t dst = make_rsrc(p.save + plane * p.plane_stride + it * p.iter_stride + grow0 * 128, row_bytes);
    if (BF && plane < 5) {
      const unsigned ho = (lane >> 5) * 512 + (lane & 31) * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) buf_st4_bf16(dst, ho + j * 1024, ld4(r_lane + 2 * j * LDH));
      return;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) buf_st4(dst, rl_off + j * 1024, ld4(r_lane + 2 * j * LDH));
  };
  auto save_regs = [&](int plane, int it, const f32x4 (&v)[8]) {  // C-layout registers -> [row][128]
    const rsrc_t dst = make_rsrc(p.save + plane * p.plane_stride + it * p.iter_stride + grow0 * 128, row_bytes);
    if (BF) {
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) buf_st1_bf16(dst, (4 * lq + r) * 512 + (16 * t + li) * 2, v[t][r]);
      return;
    }
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) buf_st1(dst, cl_off + (r * 128 + 16 * t) * 4, v[t][r]);
  };

  for (int it = 0; it < p.T; ++it) {
    if (SAVE) save_rows(0, it);  // h_in
    f32x4 z[8], acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) z[t] = xz[t];
    gemm<128, 4, false, 128>(w_z, 0, w_r, 0, a_lane, xf, ws, z);
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) z[t][r] = df_sigmoid_fast(z[t][r]);
    if (SAVE) save_regs(1, it, z);
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = xr[t];
    gemm<128, 4, false, 128>(w_r, 0, w_q, 0, a_lane, xf, ws, acc);
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t][r] = df_sigmoid_fast(acc[t][r]);
    if (SAVE) save_regs(2, it, acc);
    // the wave's A region is private and the GEMM's closing barrier is behind us: overwrite h with r * h
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) c_lane[r * LDH + 16 * t] = acc[t][r] * h[t][r];
    __syncthreads();
    if (SAVE) save_rows(4, it);  // r * h
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = xq[t];
    const bool last = it + 1 == p.T;
    if (!last) gemm<128, 4, false, 128>(w_q, 0, w_z, 0, a_lane, xf, ws, acc);
    else gemm<128, 4, false, 32>(w_q, 0, p.w.w_1, 0, a_lane, xf, ws, acc);
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
  // ---- MLP head: hid = W1 [h_T | x] + b1 -------------------------------------------------------------------------
  f32x4 hid[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float bia = p.w.b_1[16 * t + li];
    hid[t] = f32x4{bia, bia, bia, bia};
  }
  gemm<32, 4, false, 32>(p.w.w_1, 0, p.w.w_1, 4, a_lane, xf, ws, hid);
  gemm<32, 2, true, 32>(p.w.w_1, 4, nullptr, 0, a_lane, xf, ws, hid);
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
int df_launch_gru_fwd3(df_img before, df_img after, const int32_t* coords, const float* offs, const int32_t* counts,
                       int B, int N, int num_iters, df_gru_weights wts, float* flow, float* save, int mfma_bf16, void* stream) {
  Gru3Params p;
  p.before = before; p.after = after; p.coords = coords; p.offs = offs; p.counts = counts;
  p.N = N; p.T = num_iters; p.w = wts; p.flow = flow; p.save = save;
  p.iter_stride = (int64_t)B * N * 128;
  p.plane_stride = p.iter_stride * num_iters;
  const dim3 grid((N + 63) / 64, B);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (mfma_bf16 == 3) {
    if (save) hipLaunchKernelGGL((gru_fwd3_kernel<true, false, false, true>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((gru_fwd3_kernel<false, false, false, true>), grid, dim3(256), 0, s, p);
  } else if (save && mfma_bf16 == 2) hipLaunchKernelGGL((gru_fwd3_kernel<true, true, true>), grid, dim3(256), 0, s, p);
  else if (mfma_bf16 == 2) hipLaunchKernelGGL((gru_fwd3_kernel<false, true, true>), grid, dim3(256), 0, s, p);
  else if (save && mfma_bf16) hipLaunchKernelGGL((gru_fwd3_kernel<true, true>), grid, dim3(256), 0, s, p);
  else if (save) hipLaunchKernelGGL((gru_fwd3_kernel<true, false>), grid, dim3(256), 0, s, p);
  else if (mfma_bf16) hipLaunchKernelGGL((gru_fwd3_kernel<false, true>), grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL((gru_fwd3_kernel<false, false>), grid, dim3(256), 0, s, p);
  DF_CHECK_LAUNCH();
  return DF_OK;
}
