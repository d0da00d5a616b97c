// Backward of the ConvGRU decoder ([REF decoder.py:123-183] differentiated), two workgroups per CU; the forward is
// decoder3.hip, the first generation (one workgroup per CU, DF_GRU_V1=1) is decoder_bwd.hip.
//
//   * dh is accumulated IN PLACE: dh <- dh (1 - z), then the transposed-weight GEMMs add W_z^T dz_pre, r * (W_q^T dq_pre)
//     and W_r^T dr_pre into the same registers -- at most four 32-register planes are live at any point;
//   * GEMM order per step: dz_pre (needs only h, z, q), dq_pre, then dr_pre -- the A region (16 x 128) holds one gate
//     gradient at a time and each is copied to its saved plane (coalesced rows) while it sits there;
//   * every [192 -> 128 | 64] output is two GEMMs over the same A operand (128 weight rows, then 64): 128-row B tiles,
//     streamed L2 -> LDS by DMA (gemm_dma.h);
//   * saved planes are read straight into MFMA C-layout registers with buffer loads whose byte range is the bounds
//     check (rows past the sample's count read 0, stores to them are dropped): no per-row predicates;
//   * the A region is wave-private, so its write -> read hand-offs need only wave-level ordering (LDS executes one
//     wave's instructions in order); workgroup barriers remain only around the shared weight buffers.
#include "common.h"
#include "gemm_dma.h"

namespace {

using namespace gd;

__device__ __forceinline__ f32x4 buf_ld4(rsrc_t r, unsigned voff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
}
__device__ __forceinline__ float buf_ld1(rsrc_t r, unsigned voff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0));
}

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct GruBwd3Params {
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

// W16 (with BF): p.w.w_1 and p.wt.{wt_zr, wt_q, wt_1} point at bf16 copies (gemm_dma.h, WStreamT<2>)
// X2 (fp32 training, mfma_bf16 == 3): the weight pointers hold the pre-split two-plane rows of WStreamT<3>; planes stay fp32
template <bool BF, bool W16 = false, bool X2 = false>
__global__ __launch_bounds__(256, 2) void gru_bwd3_kernel(GruBwd3Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(16))) float Bs[2 * BT];
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
  static_assert(!X2 || (!BF && !W16), "X2 is a mode of its own");
  constexpr bool K8 = W16 || X2;
  constexpr int WSC = W16 ? 2 : 1;                   // weight ROW offsets in floats: halved for bf16 data (X2 rows keep the fp32 pitch)
  constexpr int CSC = K8 ? 2 : 1;                    // weight COLUMN offsets in floats: halved for 16-bit elements
  const float* a_lane = Aw + li * LDH + lq * (K8 ? 8 : 4);
  float* c_lane = Aw + 4 * lq * LDH + li;            // C-layout element (row 4 lq + r, col 16 t + li) = c_lane[r * LDH + 16 t]
  float* r_lane = Aw + (lane >> 5) * LDH + (lane & 31) * 4;  // row copies: float4 j at r_lane + 2 j LDH
  const unsigned nvalid = (unsigned)min(max(cnt - wp0, 0), 16);  // valid rows of this wave's 16-row tile
  const unsigned row_bytes = nvalid * 512u;
  const unsigned rl_off = ((lane >> 5) * 128 + (lane & 31) * 4) * 4, cl_off = (4 * lq * 128 + li) * 4;
  const float* wt_q = p.wt.wt_q;
  const float* wt_zr = p.wt.wt_zr;

  WStreamT<X2 ? 3 : BF ? (W16 ? 2 : 1) : 0> ws;
  wstream_init(ws, Bs);
  dma_first<32, 192>(p.w.w_1, 0, Bs, ws);

  auto lds_to_rows = [&](float* dst) {  // the wave's 16 x 128 A region -> global rows (coalesced; invalid rows dropped)
    const rsrc_t d = make_rsrc(dst + grow0 * 128, row_bytes);
#pragma unroll
    for (int j = 0; j < 8; ++j) buf_st4(d, rl_off + j * 1024, ld4(r_lane + 2 * j * LDH));
  };
  auto lds_to_plane = [&](float* dst) {  // ... into a saved plane (bf16 mode: bf16 half rows, see decoder3.hip)
    if (!BF) { lds_to_rows(dst); return; }
    const rsrc_t d = make_rsrc(dst + grow0 * 128, row_bytes);
    const unsigned ho = (lane >> 5) * 512 + (lane & 31) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) buf_st4_bf16(d, ho + j * 1024, ld4(r_lane + 2 * j * LDH));
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
    if (BF) {
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[t][r] = buf_ld1_bf16(s0, (4 * lq + r) * 512 + (16 * t + li) * 2);
      return;
    }
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[t][r] = buf_ld1(s0, cl_off + (r * 128 + 16 * t) * 4);
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
    for (int k = 0; k < 4; ++k) xf[k] = ld4(a_lane + (K8 ? (k >> 1) * 32 + (k & 1) * 4 : k * 16));
    wave_lds_sync();
  }
  {  // h_T rows -> A region
    const rsrc_t s0 = make_rsrc(p.save + 5 * p.plane_stride + grow0 * 128, row_bytes);
#pragma unroll
    for (int j = 0; j < 8; ++j) st4(r_lane + 2 * j * LDH, buf_ld4(s0, rl_off + j * 1024));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
  // pre1 = [h_T | x] W1^T + b1; the last chunk prefetches the first tile of W1^T (rows 0..127, 32 wide)
  gemm<32, 4, false, 32>(p.w.w_1, 0, p.w.w_1, 4, a_lane, xf, ws, pre1);
  gemm<32, 2, true, 128, 192, 32>(p.w.w_1, 4, p.wt.wt_1, 0, a_lane, xf, ws, pre1);
  {
    float df[4][3];
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
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pre = pre1[t][r];
        const float dhid = df[r][0] * w20 + df[r][1] * w21 + df[r][2] * w22;
        const float dp = dhid * df_gelu_grad(pre);
        const float hv = df_gelu(pre);
        buf_st1(dp1, ((4 * lq + r) * 32 + col) * 4, dp);
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
  gemm<128, 1, false, 64, 32, 32>(p.wt.wt_1, 0, p.wt.wt_1 + 128 * 32 / WSC, 0, a_lane, xf, ws, dh);
  gemm<64, 1, false, 128, 32, 256>(p.wt.wt_1 + 128 * 32 / WSC, 0, wt_zr, 0, a_lane, xf, ws, dxa);

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
    lds_to_plane(pl_z);  // dz_pre replaces z
    colsum(0);
    gemm<128, 4, false, 64, 256, 256>(wt_zr, 0, wt_zr + 128 * 256 / WSC, 0, a_lane, xf, ws, dh);
    gemm<64, 4, false, 128, 256, 128>(wt_zr + 128 * 256 / WSC, 0, wt_q, 0, a_lane, xf, ws, dxa);
    c_to_lds(q);
    wave_lds_sync();
    lds_to_plane(pl_q);  // dq_pre replaces q
    colsum(2);
    {
      f32x4 drh[8];
      c_load(q, pl_r);  // q <- r (lands during the GEMMs)
#pragma unroll
      for (int t = 0; t < 8; ++t) drh[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      gemm<128, 4, false, 64, 128, 128>(wt_q, 0, wt_q + 128 * 128 / WSC, 0, a_lane, xf, ws, drh);
      gemm<64, 4, false, 128, 128, 256>(wt_q + 128 * 128 / WSC, 0, wt_zr + 128 / CSC, 0, a_lane, xf, ws, dxa);
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
    lds_to_plane(pl_r);  // dr_pre replaces r
    colsum(1);
    gemm<128, 4, false, 64, 256, 256>(wt_zr + 128 / CSC, 0, wt_zr + 128 * 256 / WSC + 128 / CSC, 0, a_lane, xf, ws, dh);
    if (it > 0) gemm<64, 4, false, 128, 256, 256>(wt_zr + 128 * 256 / WSC + 128 / CSC, 0, wt_zr, 0, a_lane, xf, ws, dxa);
    else gemm<64, 4, false, 128, 256, 256>(wt_zr + 128 * 256 / WSC + 128 / CSC, 0, nullptr, 0, a_lane, xf, ws, dxa);
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
int df_launch_gru_bwd3(const float* dflow, const float* offs, const int32_t* counts, int B, int N, int num_iters,
                       df_gru_weights wts, df_gru_weights_t wtt, float* save, float* dh0, float* dx, float* dpre1,
                       float* xout, float* bias_partial, int mfma_bf16, void* stream) {
  GruBwd3Params p;
  p.dflow = dflow; p.offs = offs; p.counts = counts; p.N = N; p.T = num_iters; p.w = wts; p.wt = wtt; p.save = save;
  p.iter_stride = (int64_t)B * N * 128;
  p.plane_stride = p.iter_stride * num_iters;
  p.dh0 = dh0; p.dx = dx; p.dpre1 = dpre1; p.xout = xout; p.bias_partial = bias_partial;
  if (mfma_bf16 == 3) hipLaunchKernelGGL((gru_bwd3_kernel<false, false, true>), dim3((N + 63) / 64, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
  else if (mfma_bf16 == 2) hipLaunchKernelGGL((gru_bwd3_kernel<true, true>), dim3((N + 63) / 64, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
  else if (mfma_bf16) hipLaunchKernelGGL(gru_bwd3_kernel<true>, dim3((N + 63) / 64, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
  else hipLaunchKernelGGL(gru_bwd3_kernel<false>, dim3((N + 63) / 64, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
  DF_CHECK_LAUNCH();
  return DF_OK;
}
