// Point decoder: ConvGRUDecoder / LinearDecoder forward_single ([REF decoder.py:72-199]) as fused kernels.
//
// One workgroup = 64 valid pc0 points (4 waves x 16 points).  Each wave keeps the GRU state of its 16
// points on chip for the whole decode: h lives in MFMA C-layout registers, [h | x] is mirrored in a
// wave-private LDS region that is the A operand of every GEMM (v_mfma_f32_16x16x4_f32, exact f32).
// The three 192->128 gate matrices (295 KB fp32) do not fit LDS, so weight rows are streamed
// L2 -> registers -> LDS in 32-deep k chunks, double buffered, one chunk ahead (also across GEMMs),
// and shared by the four waves.  Gather is one contiguous 256-byte read per point per image (NHWC),
// instead of the reference's 128 strided 4-byte reads from NCHW [REF decoder.py:165-168].
//
// k permutation: MFMA step s of a 16-wide k group uses k = 16g + 4*(lane>>4) + s on both operands,
// so every fragment fetch is one ds_read_b128.
#include "common.h"
#include "gemm_stream.h"

namespace {

using namespace gs;

struct GruFwdParams {
  df_img before, after;
  const int32_t* coords;
  const float* offs;
  const int32_t* counts;
  int N, T;
  df_gru_weights w;
  float* flow;
  float* save;          // nullable
  int64_t plane_stride; // T * B * N * 128
  int64_t iter_stride;  // B * N * 128
};

constexpr int LDA_F = 196;  // [h(128) | x(64) | pad]

__global__ __launch_bounds__(256) void gru_fwd_kernel(GruFwdParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Bs = lds;                         // [2][256][36]
  float* As = lds + 2 * BSZ;               // [4][16][LDA_F]
  const int b = blockIdx.y;
  const int cnt = p.counts[b];
  const int p0 = blockIdx.x * 64;
  if (p0 >= cnt) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lq = lane >> 4;
  float* Aw = As + wave * 16 * LDA_F;
  const int wp0 = p0 + wave * 16;                  // first point of this wave
  const int64_t grow0 = (int64_t)b * p.N + wp0;    // global row of that point

  Stager stg;
  int par = 0;
  // prefetch the first weight chunk while the gather is in flight
  stage_load<256>(stg, p.w.w_zr, 192, 0);

  // ---- gather h0 = [before | after] at the point's pillar, and x = offset encoder -------------
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
      st4(Aw + pt * LDA_F + c4 * 4, v);
    }
    const float w0 = p.w.w_off[lane * 3 + 0], w1 = p.w.w_off[lane * 3 + 1], w2 = p.w.w_off[lane * 3 + 2];
    const float bo = p.w.b_off[lane];
    for (int pt = 0; pt < 16; ++pt) {
      float x = 0.f;
      if (wp0 + pt < cnt) {
        const float* o = p.offs + (grow0 + pt) * 3;
        x = fmaf(w2, o[2], fmaf(w1, o[1], fmaf(w0, o[0], bo)));
      }
      Aw[pt * LDA_F + 128 + lane] = x;
    }
  }
  stage_store<256>(stg, Bs);
  __syncthreads();

  f32x4 h[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) h[t][r] = Aw[(4 * lq + r) * LDA_F + 16 * t + li];

  const float* a_lane = Aw + li * LDA_F + lq * 4;
  auto save_rows = [&](int plane, int it) {  // coalesced copy of the wave's 16 x 128 block (cols 0..127 of Aw)
    float* dst = p.save + plane * p.plane_stride + it * p.iter_stride + grow0 * 128;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int f = lane + 64 * j;
      const int pt = f >> 5, c4 = f & 31;
      if (wp0 + pt < cnt) st4(dst + pt * 128 + c4 * 4, ld4(Aw + pt * LDA_F + c4 * 4));
    }
  };
  auto save_regs = [&](int plane, int it, const f32x4* v) {  // C-layout registers -> [row][128]
    float* dst = p.save + plane * p.plane_stride + it * p.iter_stride + grow0 * 128;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (wp0 + 4 * lq + r < cnt) dst[(4 * lq + r) * 128 + 16 * t + li] = v[t][r];
  };

  for (int it = 0; it < p.T; ++it) {
    if (p.save) save_rows(0, it);  // h_in
    // G1: [z | r] pre-activations
    f32x4 zr[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const float bia = p.w.b_zr[16 * t + li];
      zr[t] = f32x4{bia, bia, bia, bia};
    }
    gemm_stream<256, 128>(p.w.w_zr, 192, 6, p.w.w_q, 192, a_lane, Bs, par, zr, stg);
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) zr[t][r] = df_sigmoid_fast(zr[t][r]);
    if (p.save) {
      save_regs(1, it, zr);
      save_regs(2, it, zr + 8);
    }
    // A operand for G2: [r * h | x]
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) Aw[(4 * lq + r) * LDA_F + 16 * t + li] = zr[8 + t][r] * h[t][r];
    __syncthreads();
    if (p.save) save_rows(4, it);  // r * h
    f32x4 q[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float bia = p.w.b_q[16 * t + li];
      q[t] = f32x4{bia, bia, bia, bia};
    }
    const bool last = it + 1 == p.T;
    if (!last) gemm_stream<128, 256>(p.w.w_q, 192, 6, p.w.w_zr, 192, a_lane, Bs, par, q, stg);
    else gemm_stream<128, 32>(p.w.w_q, 192, 6, p.w.w_1, 192, a_lane, Bs, par, q, stg);
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        q[t][r] = df_tanh_fast(q[t][r]);
        h[t][r] = (1.f - zr[t][r]) * h[t][r] + zr[t][r] * q[t][r];
        Aw[(4 * lq + r) * LDA_F + 16 * t + li] = h[t][r];
      }
    if (p.save) save_regs(3, it, q);
    __syncthreads();
  }
  if (p.save) {  // h_T
    float* dst = p.save + 5 * p.plane_stride + grow0 * 128;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int f = lane + 64 * j;
      const int pt = f >> 5, c4 = f & 31;
      if (wp0 + pt < cnt) st4(dst + pt * 128 + c4 * 4, ld4(Aw + pt * LDA_F + c4 * 4));
    }
  }
  // ---- MLP head: Linear(192->32) + GELU + Linear(32->3) ---------------------------------------
  f32x4 hid[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float bia = p.w.b_1[16 * t + li];
    hid[t] = f32x4{bia, bia, bia, bia};
  }
  gemm_stream<32, 32>(p.w.w_1, 192, 6, nullptr, 0, a_lane, Bs, par, hid, stg);
  // hidden -> wave-private LDS (reuse cols 0..31 of Aw; h is no longer needed there)
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) Aw[(4 * lq + r) * LDA_F + 16 * t + li] = df_gelu(hid[t][r]);
  __syncthreads();
  if (lane < 48) {
    const int pt = lane / 3, o = lane - pt * 3;
    if (wp0 + pt < cnt) {
      float a = p.w.b_2[o];
      for (int c = 0; c < 32; ++c) a = fmaf(p.w.w_2[o * 32 + c], Aw[pt * LDA_F + c], a);
      p.flow[(grow0 + pt) * 3 + o] = a;
    }
  }
}

// ------------------------------------------------------------------------- LinearDecoder ---
// [REF decoder.py:72-120]: flow = W2 gelu(W1 [before | after | offset_enc(128)] + b1) + b2.  K = 256.
struct LinFwdParams {
  df_img before, after;
  const int32_t* coords;
  const float* offs;
  const int32_t* counts;
  int N;
  const float *w_off, *b_off, *w_1, *b_1, *w_2, *b_2;
  float* flow;
};
constexpr int LDA_L = 260;
constexpr int BSZ_L = 32 * LDB;  // B buffer of the 32-row weight tile  // 384 + 4; 97 slots of 16 B, 97 mod 16 = 1

__global__ __launch_bounds__(256) void linear_fwd_kernel(LinFwdParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Bs = lds;
  float* As = lds + 2 * BSZ_L;  // [4][16][LDA_L]
  const int b = blockIdx.y;
  const int cnt = p.counts[b];
  const int p0 = blockIdx.x * 64;
  if (p0 >= cnt) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lq = lane >> 4;
  float* Aw = As + wave * 16 * LDA_L;
  const int wp0 = p0 + wave * 16;
  const int64_t grow0 = (int64_t)b * p.N + wp0;
  Stager stg;
  int par = 0;
  stage_load<32>(stg, p.w_1, 256, 0);
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
    st4(Aw + pt * LDA_L + c4 * 4, v);
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
      Aw[pt * LDA_L + 128 + o] = x;
    }
  }
  stage_store<32>(stg, Bs);
  __syncthreads();
  f32x4 hid[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float bia = p.b_1[16 * t + li];
    hid[t] = f32x4{bia, bia, bia, bia};
  }
  gemm_stream<32, 32, BSZ_L>(p.w_1, 256, 8, nullptr, 0, Aw + li * LDA_L + lq * 4, Bs, par, hid, stg);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) Aw[(4 * lq + r) * LDA_L + 16 * t + li] = df_gelu(hid[t][r]);
  __syncthreads();
  if (lane < 48) {
    const int pt = lane / 3, o = lane - pt * 3;
    if (wp0 + pt < cnt) {
      float a = p.b_2[o];
      for (int c = 0; c < 32; ++c) a = fmaf(p.w_2[o * 32 + c], Aw[pt * LDA_L + c], a);
      p.flow[(grow0 + pt) * 3 + o] = a;
    }
  }
}

bool img64_ok(const df_img& d, int B) {
  return d.ptr && df_aligned16(d.ptr) && d.n == B && d.c == 64 && (d.ld % 4) == 0 && (d.img_stride % 4) == 0 &&
         (d.grp_off % 4) == 0 && d.grp_size > 0;
}

}  // namespace

int df_launch_gru_fwd3(df_img before, df_img after, const int32_t* coords, const float* offs, const int32_t* counts,
                       int B, int N, int num_iters, df_gru_weights wts, float* flow, float* save, int mfma_bf16, void* stream);

// w [rows][ld] fp32 -> out [rows][hi (ld) | lo (ld)] bf16, hi = bf16(w), lo = bf16(w - hi): the pre-split weight rows of the
// decoder kernels' mfma_bf16 = 3 form (gemm_dma.h, WStreamT<3>); one pass per optimizer step and weight matrix
__global__ __launch_bounds__(256) void split_bf16x2_rows_kernel(const float* __restrict__ w, __bf16* __restrict__ out, int64_t n, int ld) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / ld;
    const int c = (int)(i - r * ld);
    const float v = w[i];
    const __bf16 hi = (__bf16)v;
    out[r * 2 * ld + c] = hi;
    out[r * 2 * ld + ld + c] = (__bf16)(v - (float)hi);
  }
}

extern "C" int df_split_bf16x2_rows(const float* w, void* out, int64_t rows, int ld, void* stream) {
  DF_REQUIRE(w && out && rows > 0 && ld > 0 && (ld % 32) == 0 && df_aligned16(out), DF_E_ARG);
  const int64_t n = rows * ld;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(split_bf16x2_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w,
                     reinterpret_cast<__bf16*>(out), n, ld);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

extern "C" int df_gru_decoder_fwd(df_img before, df_img after, const int32_t* coords, const float* offs,
                                  const int32_t* counts, int B, int N, int num_iters, df_gru_weights wts, float* flow,
                                  float* save, void* stream) {
  return df_gru_decoder_fwd_mp(before, after, coords, offs, counts, B, N, num_iters, wts, flow, save, 0, stream);
}

extern "C" int df_gru_decoder_fwd_mp(df_img before, df_img after, const int32_t* coords, const float* offs,
                                     const int32_t* counts, int B, int N, int num_iters, df_gru_weights wts, float* flow,
                                     float* save, int mfma_bf16, void* stream) {
  DF_REQUIRE(img64_ok(before, B) && img64_ok(after, B), DF_E_SHAPE);
  DF_REQUIRE(before.h == after.h && before.w == after.w, DF_E_SHAPE);
  DF_REQUIRE(coords && offs && counts && flow && B > 0 && N > 0 && num_iters >= 1, DF_E_ARG);
  DF_REQUIRE(wts.w_off && wts.b_off && wts.w_zr && wts.b_zr && wts.w_q && wts.b_q && wts.w_1 && wts.b_1 && wts.w_2 &&
                 wts.b_2 && df_aligned16(wts.w_zr) && df_aligned16(wts.w_q) && df_aligned16(wts.w_1),
             DF_E_ARG);
  static const bool use_v1 = getenv("DF_GRU_V1") != nullptr;  // first-generation kernel (1 workgroup / CU), for A/B
  if (!use_v1) return df_launch_gru_fwd3(before, after, coords, offs, counts, B, N, num_iters, wts, flow, save, mfma_bf16, stream);
  GruFwdParams p;
  p.before = before; p.after = after; p.coords = coords; p.offs = offs; p.counts = counts;
  p.N = N; p.T = num_iters; p.w = wts; p.flow = flow; p.save = save;
  p.iter_stride = (int64_t)B * N * 128;
  p.plane_stride = p.iter_stride * num_iters;
  const size_t lds_bytes = (size_t)(2 * BSZ + 4 * 16 * LDA_F) * sizeof(float);
  DF_SET_LDS_ONCE((gru_fwd_kernel), (int)lds_bytes);
  hipLaunchKernelGGL(gru_fwd_kernel, dim3((N + 63) / 64, B), dim3(256), lds_bytes,
                     reinterpret_cast<hipStream_t>(stream), p);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

extern "C" int df_linear_decoder_fwd(df_img before, df_img after, const int32_t* coords, const float* offs,
                                     const int32_t* counts, int B, int N, const float* w_off, const float* b_off,
                                     const float* w_1, const float* b_1, const float* w_2, const float* b_2, float* flow,
                                     void* stream) {
  DF_REQUIRE(img64_ok(before, B) && img64_ok(after, B), DF_E_SHAPE);
  DF_REQUIRE(before.h == after.h && before.w == after.w, DF_E_SHAPE);
  DF_REQUIRE(coords && offs && counts && flow && w_off && b_off && w_1 && b_1 && w_2 && b_2 && df_aligned16(w_1), DF_E_ARG);
  LinFwdParams p;
  p.before = before; p.after = after; p.coords = coords; p.offs = offs; p.counts = counts; p.N = N;
  p.w_off = w_off; p.b_off = b_off; p.w_1 = w_1; p.b_1 = b_1; p.w_2 = w_2; p.b_2 = b_2; p.flow = flow;
  const size_t lds_bytes = (size_t)(2 * BSZ_L + 4 * 16 * LDA_L) * sizeof(float);
  DF_SET_LDS_ONCE((linear_fwd_kernel), (int)lds_bytes);
  hipLaunchKernelGGL(linear_fwd_kernel, dim3((N + 63) / 64, B), dim3(256), lds_bytes,
                     reinterpret_cast<hipStream_t>(stream), p);
  DF_CHECK_LAUNCH();
  return DF_OK;
}
