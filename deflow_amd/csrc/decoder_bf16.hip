// ConvGRU decoder forward on bf16 MFMA (inference path of BASELINE configs[4]; [REF decoder.py:123-183]).
//
// The fp32 kernel (decoder3.hip) with the three gate GEMMs and the MLP head's first layer on v_mfma_f32_16x16x32_bf16:
// weights are bf16 ([rows][192], converted once on the host side), the hidden state h stays fp32 in MFMA C-layout
// registers and is rounded to bf16 only as the A operand (wave-private LDS, 16 x 128 bf16), gate non-linearities and
// the blend are fp32, the x projection is hoisted as in the fp32 kernel.  A weight chunk is 128 rows x 64 k (128-byte
// rows, XOR slot swizzle, LDS-DMA): one ds_read_b128 per MFMA operand, two k steps per chunk -- the kernel is bound by
// streaming the weights from L2, not by the matrix pipe.  Inference only (no saved planes).
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr int BT = 128 * 32;        // one weight buffer: 128 rows x 128 bytes (floats)
constexpr int PA = 136;             // A region pitch in bf16 elements (272 bytes: 17 slots, conflict-free b128 rows)

struct WS {
  float* Bs;
  int par, wave;
  unsigned voff;        // per-lane DMA source byte offset: row (wave * 8 + lane / 8) * 384 + swizzled slot * 16
  const float* b_lane;  // row li of buffer 0
  int bsl[2];           // float offsets of this lane's slot in k step 0 / 1
};

template <int ROWS>
__device__ __forceinline__ void dma_chunk(const __bf16* W, int chunk, float* Bbuf, int wave, unsigned voff) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(W), 0, 0x7fffffffu, 0x00020000);
#pragma unroll
  for (int i = 0; i < ROWS / 32; ++i)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(Bbuf + (wave + 4 * i) * 256), 16, voff,
                                             (unsigned)((i * 32 * 192 + chunk * 64) * 2), 0, 0);
}

// acc[t] += A[16, 64 NCH] * W[ROWS, chunks c0 .. c0 + NCH)^T; A from the LDS region (XA = false: chunk c at +64 c
// elements) or from the two x fragments.  Same pipeline contract as gd::gemm (gemm_dma.h).
template <int ROWS, int NCH, bool XA, int ROWS_NEXT>
__device__ __forceinline__ void gemm16(const __bf16* W, int c0, const __bf16* Wn, int cn, const __bf16* a_lane,
                                       const bf16x8 (&xf)[2], WS& ws, f32x4 (&acc)[ROWS / 16]) {
#pragma unroll 1
  for (int c = 0; c < NCH; ++c) {
    float* nb = ws.Bs + ((ws.par + c + 1) & 1) * BT;
    if (c + 1 < NCH) dma_chunk<ROWS>(W, c0 + c + 1, nb, ws.wave, ws.voff);
    else if (Wn) dma_chunk<ROWS_NEXT>(Wn, cn, nb, ws.wave, ws.voff);
    const float* bb = ws.b_lane + ((ws.par + c) & 1) * BT;
    bf16x8 a0, a1;
    if (XA) {
      a0 = xf[0];
      a1 = xf[1];
    } else {
      a0 = *reinterpret_cast<const bf16x8*>(a_lane + 64 * c);
      a1 = *reinterpret_cast<const bf16x8*>(a_lane + 64 * c + 32);
    }
#pragma unroll
    for (int t = 0; t < ROWS / 16; ++t) {
      const f32x4 b0 = ld4(bb + t * 512 + ws.bsl[0]), b1 = ld4(bb + t * 512 + ws.bsl[1]);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, __builtin_bit_cast(bf16x8, b0), acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, __builtin_bit_cast(bf16x8, b1), acc[t], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  ws.par = (ws.par + NCH) & 1;
}

struct GruHParams {
  df_img before, after;     // fp32 images
  const int32_t* coords;
  const float* offs;
  const int32_t* counts;
  int N, T;
  const float *w_off, *b_off;                 // fp32 [64,3], [64]
  const __bf16 *w_zr, *w_q, *w_1;             // bf16 [256,192], [128,192], [32,192]
  const float *b_zr, *b_q, *b_1, *w_2, *b_2;  // fp32
  float* flow;
};

__global__ __launch_bounds__(256, 2) void gru_fwd_bf16_kernel(GruHParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(16))) float Bs[2 * BT];            // 32 KB
  __shared__ __attribute__((aligned(16))) __bf16 As[4 * 16 * PA];      // 17.4 KB
  const int b = blockIdx.y;
  const int cnt = p.counts[b];
  const int p0 = blockIdx.x * 64;
  if (p0 >= cnt) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  __bf16* Aw = As + wave * 16 * PA;
  const int wp0 = p0 + wave * 16;
  const int64_t grow0 = (int64_t)b * p.N + wp0;
  const __bf16* a_lane = Aw + li * PA + 8 * lq;
  __bf16* c_lane = Aw + 4 * lq * PA + li;      // C-layout element (row 4 lq + r, col 16 t + li) = c_lane[r * PA + 16 t]
  const __bf16* w_z = p.w_zr;
  const __bf16* w_r = p.w_zr + 128 * 192;

  WS ws;
  ws.Bs = Bs; ws.par = 0; ws.wave = wave;
  {
    const int row = wave * 8 + (lane >> 3), c4 = lane & 7;
    ws.voff = (unsigned)(row * 384 + ((c4 ^ ((row >> 1) & 7)) * 16));
    ws.b_lane = Bs + li * 32;
    ws.bsl[0] = ((lq) ^ ((li >> 1) & 7)) * 4;
    ws.bsl[1] = ((4 + lq) ^ ((li >> 1) & 7)) * 4;
  }
  dma_chunk<128>(w_z, 2, Bs, wave, ws.voff);   // first chunk of the x projection (k = 128 .. 191)

  // ---- x = offset encoder -> A region (bf16, temporarily) -> two register fragments -----------------------------
  bf16x8 xf[2];
  {
    const float w0 = p.w_off[lane * 3 + 0], w1 = p.w_off[lane * 3 + 1], w2 = p.w_off[lane * 3 + 2];
    const float bo = p.b_off[lane];
    for (int pt = 0; pt < 16; ++pt) {
      float x = 0.f;
      if (wp0 + pt < cnt) {
        const float* o = p.offs + (grow0 + pt) * 3;
        x = fmaf(w2, o[2], fmaf(w1, o[1], fmaf(w0, o[0], bo)));
      }
      Aw[pt * PA + lane] = (__bf16)x;
    }
  }
  __syncthreads();
  xf[0] = *reinterpret_cast<const bf16x8*>(a_lane);
  xf[1] = *reinterpret_cast<const bf16x8*>(a_lane + 32);
  __syncthreads();
  // ---- gather h0 = [before | after] straight into C-layout registers (64-byte segments), bf16 copy into the A region --
  f32x4 h[8];
  {
    const float* bp = reinterpret_cast<const float*>(p.before.ptr) + df_img_base(p.before, b);
    const float* ap = reinterpret_cast<const float*>(p.after.ptr) + df_img_base(p.after, b);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = wp0 + 4 * lq + r < cnt;
      int64_t cell = 0;
      if (ok) {
        const int32_t* cc = p.coords + (grow0 + 4 * lq + r) * 3;
        cell = (int64_t)cc[1] * p.before.w + cc[2];
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const float v = ok ? (t < 4 ? bp[cell * p.before.ld + 16 * t + li] : ap[cell * p.after.ld + 16 * (t - 4) + li]) : 0.f;
        h[t][r] = v;
        c_lane[r * PA + 16 * t] = (__bf16)v;
      }
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
  bias_init(xz, p.b_zr);
  gemm16<128, 1, true, 128>(w_z, 2, w_r, 2, a_lane, xf, ws, xz);
  bias_init(xr, p.b_zr + 128);
  gemm16<128, 1, true, 128>(w_r, 2, p.w_q, 2, a_lane, xf, ws, xr);
  bias_init(xq, p.b_q);
  gemm16<128, 1, true, 128>(p.w_q, 2, w_z, 0, a_lane, xf, ws, xq);

  for (int it = 0; it < p.T; ++it) {
    f32x4 z[8], acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) z[t] = xz[t];
    gemm16<128, 2, false, 128>(w_z, 0, w_r, 0, a_lane, xf, ws, z);
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) z[t][r] = df_sigmoid_fast(z[t][r]);
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = xr[t];
    gemm16<128, 2, false, 128>(w_r, 0, p.w_q, 0, a_lane, xf, ws, acc);
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) c_lane[r * PA + 16 * t] = (__bf16)(df_sigmoid_fast(acc[t][r]) * h[t][r]);   // r * h
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = xq[t];
    const bool last = it + 1 == p.T;
    if (!last) gemm16<128, 2, false, 128>(p.w_q, 0, w_z, 0, a_lane, xf, ws, acc);
    else gemm16<128, 2, false, 32>(p.w_q, 0, p.w_1, 0, a_lane, xf, ws, acc);
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float q = df_tanh_fast(acc[t][r]);
        h[t][r] = (1.f - z[t][r]) * h[t][r] + z[t][r] * q;
        c_lane[r * PA + 16 * t] = (__bf16)h[t][r];
      }
    __syncthreads();
  }
  // ---- MLP head: hid = gelu(W1 [h_T | x] + b1); flow = W2 hid + b2 ---------------------------------------------
  f32x4 hid[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float bia = p.b_1[16 * t + li];
    hid[t] = f32x4{bia, bia, bia, bia};
  }
  gemm16<32, 2, false, 32>(p.w_1, 0, p.w_1, 2, a_lane, xf, ws, hid);
  gemm16<32, 1, true, 32>(p.w_1, 2, nullptr, 0, a_lane, xf, ws, hid);
  float* hw = reinterpret_cast<float*>(Aw);   // the wave's A region, reused as fp32 [16][36]
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) hw[(4 * lq + r) * 36 + 16 * t + li] = df_gelu(hid[t][r]);
  __syncthreads();
  if (lane < 48) {
    const int pt = lane / 3, o = lane - pt * 3;
    if (wp0 + pt < cnt) {
      float a = p.b_2[o];
      for (int c = 0; c < 32; ++c) a = fmaf(p.w_2[o * 32 + c], hw[pt * 36 + c], a);
      p.flow[(grow0 + pt) * 3 + o] = a;
    }
  }
#endif
}

}  // namespace

extern "C" int df_gru_decoder_fwd_bf16(df_img before, df_img after, const int32_t* coords, const float* offs,
                                       const int32_t* counts, int B, int N, int num_iters, const float* w_off,
                                       const float* b_off, const void* w_zr, const float* b_zr, const void* w_q,
                                       const float* b_q, const void* w_1, const float* b_1, const float* w_2,
                                       const float* b_2, float* flow, void* stream) {
  DF_REQUIRE(before.ptr && after.ptr && before.n == B && after.n == B && before.c == 64 && after.c == 64 &&
                 before.h == after.h && before.w == after.w,
             DF_E_SHAPE);
  DF_REQUIRE(coords && offs && counts && flow && w_off && b_off && w_zr && b_zr && w_q && b_q && w_1 && b_1 && w_2 && b_2 &&
                 B > 0 && N > 0 && num_iters >= 1,
             DF_E_ARG);
  DF_REQUIRE(df_aligned16(w_zr) && df_aligned16(w_q) && df_aligned16(w_1), DF_E_ALIGN);
  GruHParams p;
  p.before = before; p.after = after; p.coords = coords; p.offs = offs; p.counts = counts; p.N = N; p.T = num_iters;
  p.w_off = w_off; p.b_off = b_off;
  p.w_zr = reinterpret_cast<const __bf16*>(w_zr); p.w_q = reinterpret_cast<const __bf16*>(w_q);
  p.w_1 = reinterpret_cast<const __bf16*>(w_1);
  p.b_zr = b_zr; p.b_q = b_q; p.b_1 = b_1; p.w_2 = w_2; p.b_2 = b_2; p.flow = flow;
  hipLaunchKernelGGL(gru_fwd_bf16_kernel, dim3((N + 63) / 64, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
  DF_CHECK_LAUNCH();
  return DF_OK;
}
