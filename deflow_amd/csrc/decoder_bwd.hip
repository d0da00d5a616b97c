// Backward of the point decoder ([REF decoder.py:123-199] differentiated).
//
//   gru_bwd_kernel    data gradients of MLP head + num_iters GRU steps for 64 points per workgroup; same
//                     structure as the forward (state in C-layout registers, wave-private LDS A operand,
//                     streamed TRANSPOSED weights).  Gate pre-activation gradients overwrite the saved
//                     z / r / q planes in place; the weight gradients are then plain split-K GEMMs over
//                     those planes (df_conv2d_wgrad, 1x1 mode).
//   gather_bwd_kernel the reference's backward of img[:, y, x] is an atomic scatter-add with duplicate
//                     indices; here each BEV cell sums the rows of its own points (the pillar sort already
//                     made them contiguous) -- deterministic, every gradient pixel written exactly once.
//   small_outer       tiny [na x nb] outer-product reductions over valid rows (offset encoder, last Linear,
//                     bias gradients).
#include "common.h"
#include "gemm_stream.h"

namespace {

using namespace gs;

constexpr int LDA_B = 260;          // [256 | pad] floats; 65 slots of 16 B, 65 mod 16 = 1
constexpr int BS_B = 192 * LDB;     // B buffer: 192 weight rows

struct GruBwdParams {
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
  float* bias_partial;  // [blocks][772]: per-workgroup partial sums of all small gradients (layout at the end of the kernel)
};

__global__ __launch_bounds__(256) void gru_bwd_kernel(GruBwdParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Bs = lds;                 // [2][192][36]
  float* As = lds + 2 * BS_B;      // [4][16][LDA_B]
  const int b = blockIdx.y;
  const int cnt = p.counts[b];
  const int p0 = blockIdx.x * 64;
  if (p0 >= cnt) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lq = lane >> 4;
  float* Aw = As + wave * 16 * LDA_B;
  const int wp0 = p0 + wave * 16;
  const int64_t grow0 = (int64_t)b * p.N + wp0;
  const float* a_lane = Aw + li * LDA_B + lq * 4;

  Stager stg;
  int par = 0;
  stage_load<32>(stg, p.w.w_1, 192, 0);

  // coalesced 16 x 128 block copies between global rows and LDS columns [col0, col0 + 128)
  auto rows_to_lds = [&](const float* src, int col0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int f = lane + 64 * j;
      const int pt = f >> 5, c4 = f & 31;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (wp0 + pt < cnt) v = ld4(src + (grow0 + pt) * 128 + c4 * 4);
      st4(Aw + pt * LDA_B + col0 + c4 * 4, v);
    }
  };
  auto lds_to_rows = [&](float* dst, int col0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int f = lane + 64 * j;
      const int pt = f >> 5, c4 = f & 31;
      if (wp0 + pt < cnt) st4(dst + (grow0 + pt) * 128 + c4 * 4, ld4(Aw + pt * LDA_B + col0 + c4 * 4));
    }
  };
  auto lds_to_c = [&](f32x4 (&v)[8], int col0) {
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[t][r] = Aw[(4 * lq + r) * LDA_B + col0 + 16 * t + li];
  };
  auto c_to_lds = [&](const f32x4 (&v)[8], int col0) {
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) Aw[(4 * lq + r) * LDA_B + col0 + 16 * t + li] = v[t][r];
  };

  // ---- [h_T | x] -> A region; x also to global for the weight-gradient GEMMs -------------------
  rows_to_lds(p.save + 5 * p.plane_stride, 0);
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
      Aw[pt * LDA_B + 128 + lane] = x;
    }
  }
  stage_store<32>(stg, Bs);
  __syncthreads();

  float sb[3][8], sb1[2] = {0.f, 0.f};
  float sw2[2][3], sdf[3];  // partials of dW2[o][col] = sum_rows dflow[row][o] hid[row][col] and db2[o]
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int t = 0; t < 8; ++t) sb[g][t] = 0.f;
  // ---- MLP head backward -----------------------------------------------------------------------
  f32x4 pre1[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float bia = p.w.b_1[16 * t + li];
    pre1[t] = f32x4{bia, bia, bia, bia};
  }
  gemm_stream<32, 192, BS_B>(p.w.w_1, 192, 6, p.wt.wt_1, 32, a_lane, Bs, par, pre1, stg);
  {
    float df[4][3];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int o = 0; o < 3; ++o) df[r][o] = (wp0 + 4 * lq + r < cnt) ? p.dflow[(grow0 + 4 * lq + r) * 3 + o] : 0.f;
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
        if (wp0 + 4 * lq + r < cnt) p.dpre1[(grow0 + 4 * lq + r) * 32 + col] = dp;
#pragma unroll
        for (int o = 0; o < 3; ++o) sw2[t][o] += df[r][o] * hv;  // df is 0 on invalid rows
        Aw[(4 * lq + r) * LDA_B + col] = dp;  // A operand of the next GEMM (cols 0..31)
        if (r == 0) sb1[t] = dp; else sb1[t] += dp;
      }
    }
  }
  __syncthreads();
  f32x4 dh[8], dxa[4];
  {
    f32x4 acc[12];
#pragma unroll
    for (int t = 0; t < 12; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    gemm_stream<192, 192, BS_B>(p.wt.wt_1, 32, 1, p.wt.wt_q, 128, a_lane, Bs, par, acc, stg);
#pragma unroll
    for (int t = 0; t < 8; ++t) dh[t] = acc[t];
#pragma unroll
    for (int t = 0; t < 4; ++t) dxa[t] = acc[8 + t];
  }

  // ---- GRU steps in reverse --------------------------------------------------------------------
  for (int it = p.T - 1; it >= 0; --it) {
    float* pl_h = p.save + 0 * p.plane_stride + it * p.iter_stride;
    float* pl_z = p.save + 1 * p.plane_stride + it * p.iter_stride;
    float* pl_r = p.save + 2 * p.plane_stride + it * p.iter_stride;
    float* pl_q = p.save + 3 * p.plane_stride + it * p.iter_stride;
    f32x4 h[8], z[8], r[8], q[8];
    rows_to_lds(pl_h, 0);
    rows_to_lds(pl_z, 128);
    __syncthreads();
    lds_to_c(h, 0);
    lds_to_c(z, 128);
    __syncthreads();
    rows_to_lds(pl_r, 0);
    rows_to_lds(pl_q, 128);
    __syncthreads();
    lds_to_c(r, 0);
    lds_to_c(q, 128);
    __syncthreads();
    // h' = (1 - z) h + z q
    f32x4 dzp[8], dhc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float d = dh[t][k];
        dzp[t][k] = d * (q[t][k] - h[t][k]) * z[t][k] * (1.f - z[t][k]);  // dz_pre
        dhc[t][k] = d * (1.f - z[t][k]);
        q[t][k] = d * z[t][k] * (1.f - q[t][k] * q[t][k]);                 // dq_pre (q no longer needed)
        sb[0][t] += dzp[t][k];
        sb[2][t] += q[t][k];
      }
    c_to_lds(q, 0);
    __syncthreads();
    lds_to_rows(pl_q, 0);  // dq_pre replaces q
    f32x4 acc[12];
#pragma unroll
    for (int t = 0; t < 12; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    gemm_stream<192, 192, BS_B>(p.wt.wt_q, 128, 4, p.wt.wt_zr, 256, a_lane, Bs, par, acc, stg);
#pragma unroll
    for (int t = 0; t < 4; ++t) dxa[t] += acc[8 + t];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float drh = acc[t][k];
        dhc[t][k] += drh * r[t][k];
        r[t][k] = drh * h[t][k] * r[t][k] * (1.f - r[t][k]);  // dr_pre
        sb[1][t] += r[t][k];
      }
    c_to_lds(dzp, 0);
    c_to_lds(r, 128);
    __syncthreads();
    lds_to_rows(pl_z, 0);    // dz_pre replaces z
    lds_to_rows(pl_r, 128);  // dr_pre replaces r
#pragma unroll
    for (int t = 0; t < 12; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (it > 0) gemm_stream<192, 192, BS_B>(p.wt.wt_zr, 256, 8, p.wt.wt_q, 128, a_lane, Bs, par, acc, stg);
    else gemm_stream<192, 192, BS_B>(p.wt.wt_zr, 256, 8, nullptr, 0, a_lane, Bs, par, acc, stg);
#pragma unroll
    for (int t = 0; t < 8; ++t) dh[t] = dhc[t] + acc[t];
#pragma unroll
    for (int t = 0; t < 4; ++t) dxa[t] += acc[8 + t];
  }
  // ---- outputs: dh0 [rows,128], dx [rows,64] -----------------------------------------------------
  c_to_lds(dh, 0);
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int k = 0; k < 4; ++k) Aw[(4 * lq + k) * LDA_B + 128 + 16 * t + li] = dxa[t][k];
  __syncthreads();
  lds_to_rows(p.dh0, 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int f = lane + 64 * j;
    const int pt = f >> 4, c4 = f & 15;
    if (wp0 + pt < cnt) st4(p.dx + (grow0 + pt) * 64 + c4 * 4, ld4(Aw + pt * LDA_B + 128 + c4 * 4));
  }
  // ---- per-workgroup partial sums of every small gradient (rows beyond cnt contributed exact zeros) ------
  // layout (PW columns): [0,384) d b_z|b_r|b_q, [384,416) d b_1, [416,608) dW_off[c][d], [608,672) d b_off[c],
  // [672,768) dW_2[o][col], [768,771) d b_2[o]
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
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int o = 0; o < 3; ++o) {
        float v = sw2[t][o];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (lq == 0) red[wave * PW + 672 + o * 32 + 16 * t + li] = v;
      }
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      float v = sdf[o];  // identical on the 16 lanes of a row group: reduce over the 4 row groups only
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (lane == 0) red[wave * PW + 768 + o] = v;
    }
    if (lane == 0) red[wave * PW + 771] = 0.f;
  }
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      float v = sb[g][t];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (lq == 0) red[wave * PW + g * 128 + 16 * t + li] = v;
    }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    float v = sb1[t];
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    if (lq == 0) red[wave * PW + 384 + 16 * t + li] = v;
  }
  __syncthreads();
  for (int o = tid; o < PW; o += 256)
    p.bias_partial[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * PW + o] =
        red[o] + red[PW + o] + red[2 * PW + o] + red[3 * PW + o];
}

// ------------------------------------------------------------------------------ LinearDecoder bwd ---
// [REF decoder.py:72-120] differentiated: flow = W2 gelu(W1 [before | after | offset_enc(128)] + b1) + b2.  Recomputes the
// gather and the hidden layer (cheaper than saving them), writes [v | xenc] rows for the W1 weight-gradient GEMM.
struct LinBwdParams {
  df_img before, after;
  const int32_t* coords;
  const float* offs;
  const int32_t* counts;
  const float* dflow;
  int N;
  const float *w_off, *b_off, *w_1, *b_1, *w_2, *wt_1;  // wt_1 = W1^T [256,32]
  float *vx, *dh0, *dxe, *dpre1, *hid;
};
constexpr int BS_L = 256 * LDB;

__global__ __launch_bounds__(256) void linear_bwd_kernel(LinBwdParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Bs = lds;                 // [2][256][36]
  float* As = lds + 2 * BS_L;      // [4][16][LDA_B]
  const int b = blockIdx.y;
  const int cnt = p.counts[b];
  const int p0 = blockIdx.x * 64;
  if (p0 >= cnt) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lq = lane >> 4;
  float* Aw = As + wave * 16 * LDA_B;
  const int wp0 = p0 + wave * 16;
  const int64_t grow0 = (int64_t)b * p.N + wp0;
  const float* a_lane = Aw + li * LDA_B + lq * 4;
  Stager stg;
  int par = 0;
  stage_load<32>(stg, p.w_1, 256, 0);
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
      st4(Aw + pt * LDA_B + c4 * 4, v);
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int o = lane + 64 * half;
      const float w0 = p.w_off[o * 3 + 0], w1 = p.w_off[o * 3 + 1], w2 = p.w_off[o * 3 + 2], bo = p.b_off[o];
      for (int pt = 0; pt < 16; ++pt) {
        float x = 0.f;
        if (wp0 + pt < cnt) {
          const float* of = p.offs + (grow0 + pt) * 3;
          x = fmaf(w2, of[2], fmaf(w1, of[1], fmaf(w0, of[0], bo)));
        }
        Aw[pt * LDA_B + 128 + o] = x;
      }
    }
  }
  stage_store<32>(stg, Bs);
  __syncthreads();
  // [v | xenc] rows -> global (coalesced) for dW1 = dpre1^T [v | xenc]
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int f = lane + 64 * j;
    const int pt = f >> 6, c4 = f & 63;
    if (wp0 + pt < cnt) st4(p.vx + (grow0 + pt) * 256 + c4 * 4, ld4(Aw + pt * LDA_B + c4 * 4));
  }
  f32x4 pre1[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float bia = p.b_1[16 * t + li];
    pre1[t] = f32x4{bia, bia, bia, bia};
  }
  gemm_stream<32, 256, BS_L>(p.w_1, 256, 8, p.wt_1, 32, a_lane, Bs, par, pre1, stg);
  {
    float df[4][3];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int o = 0; o < 3; ++o) df[r][o] = (wp0 + 4 * lq + r < cnt) ? p.dflow[(grow0 + 4 * lq + r) * 3 + o] : 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int col = 16 * t + li;
      const float w20 = p.w_2[0 * 32 + col], w21 = p.w_2[1 * 32 + col], w22 = p.w_2[2 * 32 + col];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pre = pre1[t][r];
        const float dp = (df[r][0] * w20 + df[r][1] * w21 + df[r][2] * w22) * df_gelu_grad(pre);
        if (wp0 + 4 * lq + r < cnt) {
          p.hid[(grow0 + 4 * lq + r) * 32 + col] = df_gelu(pre);
          p.dpre1[(grow0 + 4 * lq + r) * 32 + col] = dp;
        }
        Aw[(4 * lq + r) * LDA_B + col] = dp;
      }
    }
  }
  __syncthreads();
  f32x4 acc[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  gemm_stream<256, 32, BS_L>(p.wt_1, 32, 1, nullptr, 0, a_lane, Bs, par, acc, stg);
  // d[v | xenc] -> LDS (C layout -> rows) -> global
#pragma unroll
  for (int t = 0; t < 16; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) Aw[(4 * lq + r) * LDA_B + 16 * t + li] = acc[t][r];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int f = lane + 64 * j;
    const int pt = f >> 5, c4 = f & 31;
    if (wp0 + pt < cnt) {
      st4(p.dh0 + (grow0 + pt) * 128 + c4 * 4, ld4(Aw + pt * LDA_B + c4 * 4));
      st4(p.dxe + (grow0 + pt) * 128 + c4 * 4, ld4(Aw + pt * LDA_B + 128 + c4 * 4));
    }
  }
}

// ------------------------------------------------------------------------------ gather bwd ---
__global__ __launch_bounds__(256) void gather_bwd_kernel(const float* __restrict__ dh0,
                                                         const uint32_t* __restrict__ idx_sorted,
                                                         const int32_t* __restrict__ cell_rng,
                                                         const int32_t* __restrict__ cpos, int N, int ncell,
                                                         df_img dbefore, df_img dafter, int acc_before, int acc_after) {
  const int b = blockIdx.y, sub = threadIdx.x & 31, grp = threadIdx.x >> 5;
  float* __restrict__ bp = reinterpret_cast<float*>(dbefore.ptr) + df_img_base(dbefore, b);
  float* __restrict__ ap = reinterpret_cast<float*>(dafter.ptr) + df_img_base(dafter, b);
  for (int cell = blockIdx.x * 8 + grp; cell < ncell; cell += gridDim.x * 8) {
    const int s = cell_rng[2 * ((int64_t)b * ncell + cell)], e = cell_rng[2 * ((int64_t)b * ncell + cell) + 1];
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    for (int i = s; i < e; ++i) {
      const uint32_t flat = idx_sorted[i];  // b * N + n
      const int cp = cpos[flat];
      a += ld4(dh0 + ((int64_t)b * N + cp) * 128 + sub * 4);
    }
    if (sub < 16) {
      if (dbefore.ptr) {   // NULL: the caller takes the `before` gradient sparsely (df_pillar_input_grad)
        float* o = bp + (int64_t)cell * dbefore.ld + sub * 4;
        if (acc_before) a += ld4(o);
        st4(o, a);
      }
    } else {
      float* o = ap + (int64_t)cell * dafter.ld + (sub - 16) * 4;
      if (acc_after) a += ld4(o);
      st4(o, a);
    }
  }
}

// ------------------------------------------------------------------------------ small outer ---
// partial[blk][i*nb + j] = sum over the block's valid rows of a[row][i] * (b ? b[row][j] : 1)
__global__ __launch_bounds__(256) void small_outer_kernel(const float* __restrict__ a, int lda, int na,
                                                          const float* __restrict__ bmat, int ldb, int nb,
                                                          const int32_t* __restrict__ counts, int rows_per_seg, int nseg,
                                                          int64_t rows, int64_t rows_per_blk, float* __restrict__ partial) {
  const int nout = na * nb;
  const int64_t r_begin = (int64_t)blockIdx.x * rows_per_blk;
  const int64_t r_end = min(r_begin + rows_per_blk, rows);
  // threads beyond nout split the rows further: lanes = 256 / nout row lanes (>= 1)
  const int row_lanes = max(256 / nout, 1);
  const int o = threadIdx.x % nout, rl = threadIdx.x / nout;
  __shared__ float red[256];
  float acc = 0.f;
  if (rl < row_lanes) {
    const int i = o / nb, j = o - i * nb;
    for (int64_t r = r_begin + rl; r < r_end; r += row_lanes) {
      const int64_t segi = r / rows_per_seg;
      const int within = (int)(r - segi * rows_per_seg);
      if (within >= counts[(int)(segi % nseg)]) continue;
      const float av = a[r * lda + i];
      acc += bmat ? av * bmat[r * ldb + j] : av;
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < nout) {
    float s = 0.f;
    for (int k = 0; k < row_lanes; ++k) s += red[k * nout + threadIdx.x];
    partial[(int64_t)blockIdx.x * nout + threadIdx.x] = s;
  }
}

}  // namespace

int df_launch_gru_bwd3(const float* dflow, const float* offs, const int32_t* counts, int B, int N, int num_iters,
                       df_gru_weights wts, df_gru_weights_t wtt, float* save, float* dh0, float* dx, float* dpre1,
                       float* xout, float* bias_partial, int mfma_bf16, void* stream);

extern "C" int df_gru_decoder_bwd(const float* dflow, const float* offs, const int32_t* counts, int B, int N,
                                  int num_iters, df_gru_weights wts, df_gru_weights_t wtt, float* save, float* dh0,
                                  float* dx, float* dpre1, float* xout, float* bias_partial, void* stream) {
  return df_gru_decoder_bwd_mp(dflow, offs, counts, B, N, num_iters, wts, wtt, save, dh0, dx, dpre1, xout, bias_partial, 0, stream);
}

extern "C" int df_gru_decoder_bwd_mp(const float* dflow, const float* offs, const int32_t* counts, int B, int N,
                                     int num_iters, df_gru_weights wts, df_gru_weights_t wtt, float* save, float* dh0,
                                     float* dx, float* dpre1, float* xout, float* bias_partial, int mfma_bf16, void* stream) {
  DF_REQUIRE(dflow && offs && counts && save && dh0 && dx && dpre1 && xout && bias_partial && B > 0 && N > 0 &&
                 num_iters >= 1,
             DF_E_ARG);
  DF_REQUIRE(wts.w_off && wts.b_off && wts.w_1 && wts.b_1 && wts.w_2 && wtt.wt_zr && wtt.wt_q && wtt.wt_1, DF_E_ARG);
  DF_REQUIRE(df_aligned16(wts.w_1) && df_aligned16(wtt.wt_zr) && df_aligned16(wtt.wt_q) && df_aligned16(wtt.wt_1) &&
                 df_aligned16(save) && df_aligned16(dh0) && df_aligned16(dx),
             DF_E_ALIGN);
  static const bool use_v1 = getenv("DF_GRU_V1") != nullptr;  // first-generation kernel (1 workgroup / CU), for A/B
  if (!use_v1)
    return df_launch_gru_bwd3(dflow, offs, counts, B, N, num_iters, wts, wtt, save, dh0, dx, dpre1, xout, bias_partial, mfma_bf16, stream);
  GruBwdParams p;
  p.dflow = dflow; p.offs = offs; p.counts = counts; p.N = N; p.T = num_iters; p.w = wts; p.wt = wtt; p.save = save;
  p.iter_stride = (int64_t)B * N * 128;
  p.plane_stride = p.iter_stride * num_iters;
  p.dh0 = dh0; p.dx = dx; p.dpre1 = dpre1; p.xout = xout; p.bias_partial = bias_partial;
  const size_t lds_bytes = (size_t)(2 * BS_B + 4 * 16 * LDA_B) * sizeof(float);
  DF_SET_LDS_ONCE((gru_bwd_kernel), (int)lds_bytes);
  hipLaunchKernelGGL(gru_bwd_kernel, dim3((N + 63) / 64, B), dim3(256), lds_bytes,
                     reinterpret_cast<hipStream_t>(stream), p);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

// Batched form (round 4, second session).  The kernel above walks its cells one at a time through FOUR dependent global round
// trips each (cell range -> point index -> compact position -> dh0 row): 0.74 ms per step at B = 16 for 2.8 GB.  Here a lane
// group (32 lanes: `before` | `after` halves, or 16 lanes when only d(after) is wanted) takes U cells per pass with all their
// ranges, then all their j-th indices, positions and rows in flight together (buffer loads: an inactive slot's offset is out of
// range and reads 0 -- no branches, no waits at merges); per cell the rows are added in the order of the kernel above
// (bit-identical).
template <int U, bool BOTH>
__global__ __launch_bounds__(256) void gather_bwd_batched_kernel(const float* __restrict__ dh0,
                                                                 const uint32_t* __restrict__ idx_sorted,
                                                                 const int32_t* __restrict__ cell_rng,
                                                                 const int32_t* __restrict__ cpos, int N, int ncell, int BN,
                                                                 df_img dbefore, df_img dafter, int acc_before, int acc_after,
                                                                 unsigned* __restrict__ amax_after) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int LPC = BOTH ? 32 : 16, CPP = 256 / LPC;
  float amx = 0.f;      // max |d(after)| this lane wrote (df_gather_bwd_m: the bound of dv an fp16x2 consumer scales by)      // lanes per cell, cells per pass of the workgroup
  const int b = blockIdx.y, sub = threadIdx.x & (LPC - 1), grp = threadIdx.x / LPC;
  const bool is_after = !BOTH || sub >= 16;
  const df_img& im = is_after ? dafter : dbefore;
  float* __restrict__ op = reinterpret_cast<float*>(im.ptr) + df_img_base(im, b) + (sub & 15) * 4;
  const int ld = im.ld;
  const bool acc = is_after ? acc_after != 0 : acc_before != 0;
  constexpr unsigned OOB = 0xF0000000u;
  const __amdgpu_buffer_rsrc_t rng_r = __builtin_amdgcn_make_buffer_rsrc((void*)(cell_rng + 2 * (int64_t)b * ncell), 0, (unsigned)ncell * 8u, 0x00020000);
  const __amdgpu_buffer_rsrc_t idx_r = __builtin_amdgcn_make_buffer_rsrc((void*)idx_sorted, 0, (unsigned)BN * 4u, 0x00020000);
  const __amdgpu_buffer_rsrc_t cpos_r = __builtin_amdgcn_make_buffer_rsrc((void*)cpos, 0, (unsigned)BN * 4u, 0x00020000);
  const __amdgpu_buffer_rsrc_t dh_r = __builtin_amdgcn_make_buffer_rsrc((void*)(dh0 + (int64_t)b * N * 128), 0, (unsigned)N * 512u, 0x00020000);
  const unsigned col = BOTH ? (unsigned)sub * 16u : 256u + (unsigned)sub * 16u;   // byte offset inside a dh0 row
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  for (int cell0 = blockIdx.x * CPP * U; cell0 < ncell; cell0 += gridDim.x * CPP * U) {
    int s[U], len[U];
    f32x4 a[U];
    int maxlen = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int cell = cell0 + grp + CPP * u;
      const u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(rng_r, cell < ncell ? (unsigned)cell * 8u : OOB, 0, 0);
      s[u] = (int)r[0];
      len[u] = (int)r[1] - (int)r[0];
      a[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < U; ++u) maxlen = max(maxlen, len[u]);
    for (int j = 0; __any(j < maxlen); ++j) {
      unsigned flat[U], cp[U];
#pragma unroll
      for (int u = 0; u < U; ++u) flat[u] = __builtin_amdgcn_raw_buffer_load_b32(idx_r, j < len[u] ? (unsigned)(s[u] + j) * 4u : OOB, 0, 0);
#pragma unroll
      for (int u = 0; u < U; ++u) cp[u] = __builtin_amdgcn_raw_buffer_load_b32(cpos_r, j < len[u] ? flat[u] * 4u : OOB, 0, 0);
#pragma unroll
      for (int u = 0; u < U; ++u)
        a[u] += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dh_r, j < len[u] ? cp[u] * 512u + col : OOB, 0, 0));
    }
    if (acc) {   // (not the engine's case: both images are written fresh there)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int cell = cell0 + grp + CPP * u;
        if (cell < ncell) a[u] += ld4(op + (int64_t)cell * ld);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int cell = cell0 + grp + CPP * u;
      if (cell < ncell) {
        st4(op + (int64_t)cell * ld, a[u]);
        if (is_after) amx = fmaxf(amx, fmaxf(fmaxf(fabsf(a[u][0]), fabsf(a[u][1])), fmaxf(fabsf(a[u][2]), fabsf(a[u][3]))));
      }
    }
  }
  if (amax_after) {     // one atomic per wavefront
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amx = fmaxf(amx, __shfl_xor(amx, o, 64));
    // (compare first: half a million wavefronts hitting one address with an unconditional atomic cost 3 ms; after the first few the
    //  load alone answers)
    const unsigned mb = __builtin_bit_cast(unsigned, amx);
    if ((threadIdx.x & 63) == 0 && mb > __atomic_load_n(amax_after, __ATOMIC_RELAXED)) atomicMax(amax_after, mb);
  }
#endif
}

static int gather_bwd_impl(const float* dh0, const uint32_t* idx_sorted, const int32_t* cell_rng, const int32_t* cpos, int B, int N,
                           df_img dbefore, df_img dafter, int accumulate_before, int accumulate_after, int nblk, float* amax_after,
                           void* stream);

extern "C" int df_gather_bwd(const float* dh0, const uint32_t* idx_sorted, const int32_t* cell_rng, const int32_t* cpos,
                             int B, int N, df_img dbefore, df_img dafter, int accumulate_before, int accumulate_after,
                             int nblk, void* stream) {
  return gather_bwd_impl(dh0, idx_sorted, cell_rng, cpos, B, N, dbefore, dafter, accumulate_before, accumulate_after, nblk, nullptr, stream);
}

// the MEASURING form (round 5): amax_after (a zeroed device scalar) receives max |d(after)| of what this call writes -- the step's
// backward asked for that bound with a df_absmax pass over the 1 GB image (0.25 ms at B = 16).  Not with accumulate_after.
extern "C" int df_gather_bwd_m(const float* dh0, const uint32_t* idx_sorted, const int32_t* cell_rng, const int32_t* cpos,
                               int B, int N, df_img dbefore, df_img dafter, int accumulate_before, int nblk, float* amax_after,
                               void* stream) {
  DF_REQUIRE(amax_after != nullptr, DF_E_ARG);
  return gather_bwd_impl(dh0, idx_sorted, cell_rng, cpos, B, N, dbefore, dafter, accumulate_before, 0, nblk, amax_after, stream);
}

static int gather_bwd_impl(const float* dh0, const uint32_t* idx_sorted, const int32_t* cell_rng, const int32_t* cpos, int B, int N,
                           df_img dbefore, df_img dafter, int accumulate_before, int accumulate_after, int nblk, float* amax_after,
                           void* stream) {
  DF_REQUIRE(dh0 && idx_sorted && cell_rng && cpos && dafter.ptr && B > 0 && N > 0 && nblk > 0, DF_E_ARG);
  DF_REQUIRE(dafter.n == B && dafter.c == 64 && (dafter.ld % 4) == 0, DF_E_SHAPE);
  if (dbefore.ptr)
    DF_REQUIRE(dbefore.n == B && dbefore.c == 64 && dbefore.h == dafter.h && dbefore.w == dafter.w && (dbefore.ld % 4) == 0,
               DF_E_SHAPE);
  static const int batched = getenv("DF_GATHER_BWD_V1") ? 0 : 1;   // A/B: the one-cell-at-a-time kernel
  DF_REQUIRE(!amax_after || (batched && (int64_t)B * N < (1 << 29) && (int64_t)dafter.h * dafter.w < (1 << 28)), DF_E_SHAPE);   // (the batched kernel measures)
  const int64_t ncell = (int64_t)dafter.h * dafter.w;
  if (batched && (int64_t)B * N < (1 << 29) && ncell < (1 << 28)) {   // (32-bit byte offsets of the buffer loads)
    constexpr int U = 4;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dbefore.ptr) {
      const int nb = (int)min((int64_t)8192, (ncell + 8 * U - 1) / (8 * U));
      hipLaunchKernelGGL((gather_bwd_batched_kernel<U, true>), dim3(nb, B), dim3(256), 0, st, dh0, idx_sorted, cell_rng, cpos, N,
                         (int)ncell, B * N, dbefore, dafter, accumulate_before, accumulate_after, reinterpret_cast<unsigned*>(amax_after));
    } else {
      const int nb = (int)min((int64_t)8192, (ncell + 16 * U - 1) / (16 * U));
      hipLaunchKernelGGL((gather_bwd_batched_kernel<U, false>), dim3(nb, B), dim3(256), 0, st, dh0, idx_sorted, cell_rng, cpos, N,
                         (int)ncell, B * N, dbefore, dafter, accumulate_before, accumulate_after, reinterpret_cast<unsigned*>(amax_after));
    }
    DF_CHECK_LAUNCH();
    return DF_OK;
  }
  hipLaunchKernelGGL(gather_bwd_kernel, dim3(nblk, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dh0,
                     idx_sorted, cell_rng, cpos, N, dafter.h * dafter.w, dbefore, dafter, accumulate_before,
                     accumulate_after);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

extern "C" int df_small_outer(const float* a, int lda, int na, const float* b, int ldb, int nb, const int32_t* counts,
                              int rows_per_seg, int nseg, int64_t rows, float* partial, int nblk, void* stream) {
  DF_REQUIRE(a && counts && partial && na > 0 && nb > 0 && na * nb <= 256 && rows > 0 && nblk > 0 && rows_per_seg > 0 &&
                 nseg > 0,
             DF_E_ARG);
  const int64_t rpb = (rows + nblk - 1) / nblk;
  hipLaunchKernelGGL(small_outer_kernel, dim3(nblk), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a, lda, na, b,
                     ldb, nb, counts, rows_per_seg, nseg, rows, rpb, partial);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

extern "C" int df_linear_decoder_bwd(df_img before, df_img after, const int32_t* coords, const float* offs,
                                     const int32_t* counts, const float* dflow, int B, int N, const float* w_off,
                                     const float* b_off, const float* w_1, const float* b_1, const float* w_2,
                                     const float* wt_1, float* vx, float* dh0, float* dxe, float* dpre1, float* hid,
                                     void* stream) {
  DF_REQUIRE(before.ptr && after.ptr && coords && offs && counts && dflow && w_off && b_off && w_1 && b_1 && w_2 && wt_1 &&
                 vx && dh0 && dxe && dpre1 && hid && B > 0 && N > 0,
             DF_E_ARG);
  DF_REQUIRE(before.n == B && after.n == B && before.c == 64 && after.c == 64 && before.h == after.h && before.w == after.w,
             DF_E_SHAPE);
  DF_REQUIRE(df_aligned16(w_1) && df_aligned16(wt_1) && df_aligned16(vx) && df_aligned16(dh0) && df_aligned16(dxe), DF_E_ALIGN);
  LinBwdParams p;
  p.before = before; p.after = after; p.coords = coords; p.offs = offs; p.counts = counts; p.dflow = dflow; p.N = N;
  p.w_off = w_off; p.b_off = b_off; p.w_1 = w_1; p.b_1 = b_1; p.w_2 = w_2; p.wt_1 = wt_1;
  p.vx = vx; p.dh0 = dh0; p.dxe = dxe; p.dpre1 = dpre1; p.hid = hid;
  const size_t lds_bytes = (size_t)(2 * BS_L + 4 * 16 * LDA_B) * sizeof(float);
  DF_SET_LDS_ONCE((linear_bwd_kernel), (int)lds_bytes);
  hipLaunchKernelGGL(linear_bwd_kernel, dim3((N + 63) / 64, B), dim3(256), lds_bytes,
                     reinterpret_cast<hipStream_t>(stream), p);
  DF_CHECK_LAUNCH();
  return DF_OK;
}
