// Weight gradients of the BEV convolutions (split from conv.hip in round 6; kernels unchanged): dw[co, tap, ci] = sum_m dy[m, co] * x[pix(m, tap), ci]
// for every kernel size / stride / storage form of the UNet ([REF decoder.py:202-220] differentiated w.r.t. the Conv2d weights),
// the split-K reductions and the weight transpose.
#include "conv_common.h"

namespace {

// ------------------------------------------------------------------------------- wgrad ---
struct WgradParams {
  df_img x, dy;
  float* ws;  // [splits][N][taps][K]
  const int32_t* row_counts;  // optional (1x1 only): pixel p of a row list is valid iff p % rows_per_seg < row_counts[p / rows_per_seg]
  int rows_per_seg;
  int stride, pad, K, N, chunks_per_row, total_chunks, chunks_per_split;
  unsigned x_bytes, dy_bytes;  // DMA path: buffer extents (0 = use the register-staged kernels)
  float* bias_ws;              // optional [splits][N]: per-split column sums of dy (bias gradient), written by ci-tile 0
  int bf16;                    // MFMA operands rounded to bf16 (fp32 tensors, fp32 accumulation)
  int xcd_map;                 // ring kernel: all (ci, co) tiles of a split on one XCD (shared x / dy tiles hit its L2)
  const float* amax_x;         // fp16x2 form (wgrad3_x3_kernel<2>): upper bounds of max|x| / max|dy| (device scalars)
  const float* amax_dy;
};

// chunk = one output-row segment of P pixels: (image n, output row oy, first column ox0)
struct WgChunk {
  int n, oy, ox0;
};
__device__ __forceinline__ WgChunk wg_chunk(const WgradParams& p, int ch, int P) {
  WgChunk c;
  const int rowid = ch / p.chunks_per_row, seg = ch - rowid * p.chunks_per_row;
  c.n = rowid / p.dy.h;
  c.oy = rowid - c.n * p.dy.h;
  c.ox0 = seg * P;
  return c;
}
__device__ __forceinline__ bool wg_row_ok(const WgradParams& p, int pix) {
  if (!p.row_counts) return true;
  const int sg = pix / p.rows_per_seg;
  return (pix - sg * p.rows_per_seg) < p.row_counts[sg];
}

// Generic k x k weight gradient: 64 co x 64 ci x all taps per workgroup.  Loads are unconditional (clamped address +
// select) and the NEXT chunk is fetched into registers while the current one is multiplied (T14).
template <int KS, int STRIDE, int P /* output pixels per chunk (one row segment) */, bool BF = false>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(WgradParams p) {
  constexpr int XW = (P - 1) * STRIDE + KS;     // input pixels needed per row
  constexpr int TAPS = KS * KS;
  constexpr int LC = 64;                        // channels per tile
  constexpr int NLY = P * (LC / 4) / 256;       // float4 loads per thread: dY tile
  constexpr int NXF = KS * XW * (LC / 4);       // float4 elements of the X patch
  constexpr int NLX = (NXF + 255) / 256;
  __shared__ __attribute__((aligned(16))) float dYs[P * LC];
  __shared__ __attribute__((aligned(16))) float Xs[KS * XW * LC];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, kh = lane >> 5;
  const int wci = wave & 1, wco = wave >> 1;
  const int ci0 = blockIdx.x * LC, co0 = blockIdx.y * LC, split = blockIdx.z;
  const bool wave_active = (ci0 + wci * 32) < p.K;  // K == 32 leaves half of the ci tile empty
  const bool do_bias = p.bias_ws && blockIdx.x == 0;
  float bsum = 0.f;

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  const float* __restrict__ xp = reinterpret_cast<const float*>(p.x.ptr);
  const float* __restrict__ dyp = reinterpret_cast<const float*>(p.dy.ptr);
  const int wy = p.dy.w, hx = p.x.h, wx = p.x.w;
  const int c_begin = split * p.chunks_per_split;
  const int c_end = min(c_begin + p.chunks_per_split, p.total_chunks);

  f32x4 ry[NLY], rx[NLX];
  unsigned my = 0, mx = 0;
  auto fetch = [&](int ch) {
    const WgChunk c = wg_chunk(p, ch, P);
    const float* yb = dyp + df_img_base(p.dy, c.n) + (int64_t)c.oy * wy * p.dy.ld + co0;
    const float* xb = xp + df_img_base(p.x, c.n) + ci0;
    my = mx = 0;
#pragma unroll
    for (int j = 0; j < NLY; ++j) {
      const int f = tid + 256 * j;
      const int px = f / (LC / 4), c4i = f - px * (LC / 4);
      const bool ok = (c.ox0 + px < wy) && wg_row_ok(p, c.ox0 + px);
      ry[j] = ld4(yb + (ok ? (c.ox0 + px) * p.dy.ld : 0) + c4i * 4);
      my |= (unsigned)ok << j;
    }
#pragma unroll
    for (int j = 0; j < NLX; ++j) {
      const int f = tid + 256 * j;
      const int c4i = f % (LC / 4);
      const int q = f / (LC / 4);
      const int xi = q % XW, ky = q / XW;
      const int iy = c.oy * STRIDE + ky - p.pad;
      const int ix = c.ox0 * STRIDE + xi - p.pad;
      const bool ok = (f < NXF) && (unsigned)iy < (unsigned)hx && (unsigned)ix < (unsigned)wx && (ci0 + c4i * 4 < p.K) &&
                      wg_row_ok(p, ix);
      rx[j] = ld4(xb + (ok ? (iy * wx + ix) * p.x.ld + c4i * 4 : -ci0));
      mx |= (unsigned)ok << j;
    }
  };
  auto stash = [&]() {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NLY; ++j) st4(dYs + (tid + 256 * j) * 4, ((my >> j) & 1) ? ry[j] : zero);
#pragma unroll
    for (int j = 0; j < NLX; ++j)
      if (tid + 256 * j < NXF) st4(Xs + (tid + 256 * j) * 4, ((mx >> j) & 1) ? rx[j] : zero);
  };

  if (c_begin < c_end) {
    fetch(c_begin);
    stash();
  }
  __syncthreads();
  for (int ch = c_begin; ch < c_end; ++ch) {
    if (ch + 1 < c_end) fetch(ch + 1);
    if (wave_active && BF) {   // bf16 operands: lane (column, kh) holds pixels 16 s + 8 kh .. + 7 of k step s
#pragma unroll
      for (int ks = 0; ks < P / 16; ++ks) {
        float av[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) av[k] = dYs[(16 * ks + 8 * kh + k) * LC + wco * 32 + li];
        const bf16x8_t a8 = pack_bf16(av);
#pragma unroll
        for (int ky = 0; ky < KS; ++ky)
#pragma unroll
          for (int kx = 0; kx < KS; ++kx) {
            float bv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) bv[k] = Xs[(ky * XW + (16 * ks + 8 * kh + k) * STRIDE + kx) * LC + wci * 32 + li];
            acc[ky * KS + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, pack_bf16(bv), acc[ky * KS + kx], 0, 0, 0);
          }
      }
    } else if (wave_active) {
#pragma unroll 4
      for (int ks = 0; ks < P / 2; ++ks) {
        const int px = 2 * ks + kh;
        const float a = dYs[px * LC + wco * 32 + li];
#pragma unroll
        for (int ky = 0; ky < KS; ++ky)
#pragma unroll
          for (int kx = 0; kx < KS; ++kx) {
            const float b = Xs[(ky * XW + px * STRIDE + kx) * LC + wci * 32 + li];
            acc[ky * KS + kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[ky * KS + kx], 0, 0, 0);
          }
      }
    }
    if (do_bias) {  // all 256 threads: column tid & 63, pixel group tid >> 6 (combined once after the loop)
#pragma unroll
      for (int j = 0; j < P / 4; ++j) bsum += dYs[((tid >> 6) * (P / 4) + j) * LC + (tid & 63)];
    }
    __syncthreads();
    if (ch + 1 < c_end) stash();
    __syncthreads();
  }
  if (do_bias) {
    dYs[tid] = bsum;
    __syncthreads();
    if (tid < LC) p.bias_ws[(int64_t)split * p.N + co0 + tid] = dYs[tid] + dYs[64 + tid] + dYs[128 + tid] + dYs[192 + tid];
  }
  if (wave_active) {
    float* o = p.ws + (int64_t)split * p.N * TAPS * p.K;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = co0 + wco * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
        const int ci = ci0 + wci * 32 + li;
        o[((int64_t)co * TAPS + t) * p.K + ci] = acc[t][e];
      }
  }
}

// 1x1 weight gradient (= row GEMM dW[co, ci] = sum_p dy[p, co] x[p, ci]) with a 128 co x CIT ci tile per workgroup:
// every operand row is read once per tile instead of once per 64x64 tile -- the decoder's weight-gradient GEMMs
// stream 2.6 GB planes and were HBM-bound on re-reads with the generic kernel.
template <int CIT, bool BF = false>
__global__ __launch_bounds__(256, 2) void wgrad1x1_kernel(WgradParams p) {
  constexpr int P = 32, COT = 128;
  constexpr int TCI = CIT / 64;               // 32-wide ci tiles per wave
  constexpr int NLY = P * (COT / 4) / 256;    // 4
  constexpr int NLX = P * (CIT / 4) / 256;    // 2 or 4
  __shared__ __attribute__((aligned(16))) float dYs[P * COT];
  __shared__ __attribute__((aligned(16))) float Xs[P * CIT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, kh = lane >> 5;
  const int wco = wave & 1, wci = wave >> 1;
  const int ci0 = blockIdx.x * CIT, co0 = blockIdx.y * COT, split = blockIdx.z;
  const int ciw = ci0 + wci * (CIT / 2);      // first ci of this wave
  const bool do_bias = p.bias_ws && blockIdx.x == 0;
  float bsum = 0.f;

  f32x16 acc[2][TCI];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TCI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const float* __restrict__ xp = reinterpret_cast<const float*>(p.x.ptr);
  const float* __restrict__ dyp = reinterpret_cast<const float*>(p.dy.ptr);
  const int wy = p.dy.w;
  const int c_begin = split * p.chunks_per_split;
  const int c_end = min(c_begin + p.chunks_per_split, p.total_chunks);

  f32x4 ry[NLY], rx[NLX];
  unsigned my = 0, mx = 0;
  auto fetch = [&](int ch) {
    const WgChunk c = wg_chunk(p, ch, P);
    const float* yb = dyp + df_img_base(p.dy, c.n) + (int64_t)c.oy * wy * p.dy.ld + co0;
    const float* xb = xp + df_img_base(p.x, c.n) + (int64_t)c.oy * wy * p.x.ld + ci0;
    my = mx = 0;
#pragma unroll
    for (int j = 0; j < NLY; ++j) {
      const int f = tid + 256 * j;
      const int px = f / (COT / 4), c4i = f - px * (COT / 4);
      const bool ok = (c.ox0 + px < wy) && wg_row_ok(p, c.ox0 + px);
      ry[j] = ld4(yb + (ok ? (c.ox0 + px) * p.dy.ld : 0) + c4i * 4);
      my |= (unsigned)ok << j;
    }
#pragma unroll
    for (int j = 0; j < NLX; ++j) {
      const int f = tid + 256 * j;
      const int px = f / (CIT / 4), c4i = f - px * (CIT / 4);
      const bool ok = (c.ox0 + px < wy) && (ci0 + c4i * 4 < p.K) && wg_row_ok(p, c.ox0 + px);
      rx[j] = ld4(xb + (ok ? (c.ox0 + px) * p.x.ld + c4i * 4 : -ci0));
      mx |= (unsigned)ok << j;
    }
  };
  auto stash = [&]() {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NLY; ++j) st4(dYs + (tid + 256 * j) * 4, ((my >> j) & 1) ? ry[j] : zero);
#pragma unroll
    for (int j = 0; j < NLX; ++j) st4(Xs + (tid + 256 * j) * 4, ((mx >> j) & 1) ? rx[j] : zero);
  };
  if (c_begin < c_end) {
    fetch(c_begin);
    stash();
  }
  __syncthreads();
  const bool wave_active = ciw < p.K;
  for (int ch = c_begin; ch < c_end; ++ch) {
    if (ch + 1 < c_end) fetch(ch + 1);
    if (wave_active && BF) {
#pragma unroll
      for (int ks = 0; ks < P / 16; ++ks) {
        bf16x8_t a8[2], b8[TCI];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          float v[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = dYs[(16 * ks + 8 * kh + k) * COT + wco * 64 + i * 32 + li];
          a8[i] = pack_bf16(v);
        }
#pragma unroll
        for (int j = 0; j < TCI; ++j) {
          float v[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = Xs[(16 * ks + 8 * kh + k) * CIT + wci * (CIT / 2) + j * 32 + li];
          b8[j] = pack_bf16(v);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < TCI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[i], b8[j], acc[i][j], 0, 0, 0);
      }
    } else if (wave_active) {
#pragma unroll 4
      for (int ks = 0; ks < P / 2; ++ks) {
        const int px = 2 * ks + kh;
        float a[2], b[TCI];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = dYs[px * COT + wco * 64 + i * 32 + li];
#pragma unroll
        for (int j = 0; j < TCI; ++j) b[j] = Xs[px * CIT + wci * (CIT / 2) + j * 32 + li];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < TCI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
    if (do_bias) {  // all 256 threads: column tid & 127, pixel half tid >> 7 (combined once after the loop)
#pragma unroll
      for (int j = 0; j < P / 2; ++j) bsum += dYs[((tid >> 7) * (P / 2) + j) * COT + (tid & 127)];
    }
    __syncthreads();
    if (ch + 1 < c_end) stash();
    __syncthreads();
  }
  if (do_bias) {
    dYs[tid] = bsum;
    __syncthreads();
    if (tid < COT) p.bias_ws[(int64_t)split * p.N + co0 + tid] = dYs[tid] + dYs[128 + tid];
  }
  if (wave_active) {
    float* o = p.ws + (int64_t)split * p.N * p.K;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < TCI; ++j) {
        const int ci = ciw + j * 32 + li;
        if (ci < p.K) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int co = co0 + wco * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
            o[(int64_t)co * p.K + ci] = acc[i][j][e];
          }
        }
      }
  }
}

// -------------------------------------------------------------------------------------------------------------
// LDS-DMA variants of the weight-gradient kernels: operand tiles go straight to double-buffered LDS with
// buffer_load_dwordx4 ... lds (out-of-range lanes = zeros = padding / masked rows), one barrier per chunk, no VGPR
// staging.  LDS images are pixel-major [pixel][channel tile] and linear (fragment reads are lane-consecutive b32).
template <int KS, int STRIDE, int P>
__global__ __launch_bounds__(256, 2) void wgrad_dma_kernel(WgradParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int XW = (P - 1) * STRIDE + KS;
  constexpr int TAPS = KS * KS;
  constexpr int LC = 64;
  constexpr int YSZ = P * LC, XSZ = ((KS * XW + 3) / 4) * 4 * LC;  // floats per buffer (X rounded up to whole DMA ops)
  constexpr int NYI = P / 4, NXI = (KS * XW + 3) / 4;               // 1-KB DMA ops (4 pixels x 64 ch) per tile
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* dYs = lds;               // [2][YSZ]
  float* Xs = lds + 2 * YSZ;      // [2][XSZ]
  typedef __attribute__((address_space(3))) void* lds_ptr_t;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kh = lane >> 5;
  const int wci = wave & 1, wco = wave >> 1;
  const int ci0 = blockIdx.x * LC, co0 = blockIdx.y * LC, split = blockIdx.z;
  const bool wave_active = (ci0 + wci * 32) < p.K;
  const bool do_bias = p.bias_ws && blockIdx.x == 0;
  float bsum = 0.f;
  const int lp = lane >> 4, lc4 = lane & 15;   // pixel within the DMA op, 16-byte channel slot

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(p.x.ptr, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.dy.ptr, 0, p.dy_bytes, 0x00020000);
  const int wy = p.dy.w, hx = p.x.h, wx = p.x.w;
  const int c_begin = split * p.chunks_per_split;
  const int c_end = min(c_begin + p.chunks_per_split, p.total_chunks);
  const bool ci_ok = ci0 + lc4 * 4 < p.K;

  auto issue = [&](int ch, int buf) {
    const WgChunk c = wg_chunk(p, ch, P);
    const int64_t yb = df_img_base(p.dy, c.n) + (int64_t)c.oy * wy * p.dy.ld + co0 + lc4 * 4;
    const int64_t xb = df_img_base(p.x, c.n) + ci0 + lc4 * 4;
    for (int k = wave; k < NYI; k += 4) {
      const int px = 4 * k + lp;
      const bool ok = c.ox0 + px < wy && wg_row_ok(p, c.ox0 + px);
      const unsigned vo = ok ? (unsigned)((yb + (int64_t)(c.ox0 + px) * p.dy.ld) * 4) : DMA_BAD;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(yr, (lds_ptr_t)(dYs + buf * YSZ + k * 256), 16, vo, 0, 0, 0);
    }
    for (int k = wave; k < NXI; k += 4) {
      const int q = 4 * k + lp;
      const int ky = q / XW, xi = q - ky * XW;
      const int iy = c.oy * STRIDE + ky - p.pad, ix = c.ox0 * STRIDE + xi - p.pad;
      const bool ok = ci_ok && ky < KS && (unsigned)iy < (unsigned)hx && (unsigned)ix < (unsigned)wx && wg_row_ok(p, ix);
      const unsigned vo = ok ? (unsigned)((xb + ((int64_t)iy * wx + ix) * p.x.ld) * 4) : DMA_BAD;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr_t)(Xs + buf * XSZ + k * 256), 16, vo, 0, 0, 0);
    }
  };

  if (c_begin < c_end) issue(c_begin, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int ch = c_begin; ch < c_end; ++ch) {
    const int buf = (ch - c_begin) & 1;
    if (ch + 1 < c_end) issue(ch + 1, buf ^ 1);
    if (wave_active) {
      const float* dyb = dYs + buf * YSZ;
      const float* xbuf = Xs + buf * XSZ;
#pragma unroll 4
      for (int ks = 0; ks < P / 2; ++ks) {
        const int px = 2 * ks + kh;
        const float a = dyb[px * LC + wco * 32 + li];
#pragma unroll
        for (int ky = 0; ky < KS; ++ky)
#pragma unroll
          for (int kx = 0; kx < KS; ++kx) {
            const float b = xbuf[(ky * XW + px * STRIDE + kx) * LC + wci * 32 + li];
            acc[ky * KS + kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[ky * KS + kx], 0, 0, 0);
          }
      }
    }
    if (do_bias) {  // all 256 threads: column tid & 63, pixel group tid >> 6 (combined once after the loop)
      const float* dyb = dYs + buf * YSZ;
#pragma unroll
      for (int j = 0; j < P / 4; ++j) bsum += dyb[((tid >> 6) * (P / 4) + j) * LC + (tid & 63)];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (do_bias) {
    dYs[tid] = bsum;
    __syncthreads();
    if (tid < LC) p.bias_ws[(int64_t)split * p.N + co0 + tid] = dYs[tid] + dYs[64 + tid] + dYs[128 + tid] + dYs[192 + tid];
  }
  if (wave_active) {
    float* o = p.ws + (int64_t)split * p.N * TAPS * p.K;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = co0 + wco * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
        const int ci = ci0 + wci * 32 + li;
        o[((int64_t)co * TAPS + t) * p.K + ci] = acc[t][e];
      }
  }
#endif
}

// 3x3 stride-1 weight gradient with 12-wave workgroups: wave = (32 co x 32 ci quadrant) x kernel row ky, 3 taps =
// 48 accumulator registers (79 VGPRs) instead of 4 waves that each hold all 9 taps (144 accumulators, 191 VGPRs, two
// waves per SIMD at most).  The three ky groups share every dY / X tile.  D = 2 (default): double-buffered, two
// workgroups per CU = six waves per SIMD -- measured +1.5..5 % over the 4-wave kernel (132 vs 126 TFLOP/s on the
// largest layers); D = 3: one workgroup per CU with a 3-deep ring -- measured 5 % SLOWER (one barrier domain per CU).
template <int P, int D, int STRIDE = 1, bool BF = false>
__global__ __launch_bounds__(768) void wgrad3_ring_kernel(WgradParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int XW = (P - 1) * STRIDE + 3, LC = 64;
  constexpr int YSZ = P * LC, NXI = (3 * XW + 3) / 4, NYI = P / 4, XSZ = NXI * 4 * LC, STG = YSZ + XSZ;
  constexpr int NOPS = NYI + NXI;                 // 1-KB DMA ops per stage (34), dealt round-robin to the 12 waves
  extern __shared__ __attribute__((aligned(16))) float lds[];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kh = lane >> 5;
  const int quad = wave & 3, ky = wave >> 2;
  const int wci = quad & 1, wco = quad >> 1;
  // The (ci, co) tiles of one split read the same x and dy tiles.  Hardware deals workgroups to the 8 XCDs round-robin in
  // linear-id order, which would put them behind 4..32 different L2s (every tile re-read from HBM: 2.7x the tensors,
  // profiles/r02_pmc_hbm_bytes.txt); the remap gives each XCD whole splits, tiles of a split dispatched back to back.
  int bx = blockIdx.x, by = blockIdx.y, split = blockIdx.z;
  if (p.xcd_map) {
    const int nt = gridDim.x * gridDim.y;
    const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int lg = df_xcd_swizzle(lin, nt * gridDim.z);
    const int tile = lg % nt;
    split = lg / nt;
    bx = tile % gridDim.x;
    by = tile / gridDim.x;
  }
  const int ci0 = bx * LC, co0 = by * LC;
  const bool wave_active = (ci0 + wci * 32) < p.K;
  const bool do_bias = p.bias_ws && bx == 0;
  float bsum = 0.f;
  const int lp = lane >> 4, lc4 = lane & 15;
  const int my_ops = (NOPS - wave + 11) / 12;     // 3 for waves 0..9, 2 for waves 10, 11

  f32x16 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(p.x.ptr, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.dy.ptr, 0, p.dy_bytes, 0x00020000);
  const int wy = p.dy.w, hx = p.x.h, wx = p.x.w;
  const int c_begin = split * p.chunks_per_split;
  const int c_end = min(c_begin + p.chunks_per_split, p.total_chunks);
  const int nst = max(c_end - c_begin, 0);
  const bool ci_ok = ci0 + lc4 * 4 < p.K;

  // DMA addressing, strength-reduced: chunks are issued in order, so a wave-uniform cursor (image, row, segment: scalar
  // registers, advanced without divisions) carries the per-stage part and every lane keeps the lane-constant part of its
  // (at most three) ops -- per stage and op one add, the border compares and a select.  (With wg_chunk()'s divisions and
  // 64-bit per-lane products redone per stage this was ~500 VALU instructions per wave and stage: invisible behind the fp32
  // MFMAs, the limiter of the bf16-operand form.)
  // Ops are dealt j = wave + 12 i: only op 0 can be a dY op (NYI <= 12), ops 1 and 2 are X ops -- one wave-uniform branch per
  // stage instead of a descriptor select per op (the scalar unit is shared by the CU's 24 waves: at 86 SALU instructions per
  // wave and stage the bf16-operand form was bound by it, profiles/r02_pmc_conv_bf16.txt).
  static_assert(NYI <= 12 && NOPS > 12, "op 0 is the only dY op; every wave has an op 1");
  const bool op0_y = wave < NYI, op2_on = wave + 24 < NOPS;
  unsigned loff[3];      // lane-constant byte offset of op i inside its tile (relative to the stage's uniform base)
  int lpx[3], lqy[3];    // dY op: pixel; X op: column offset xi - pad, row offset qy - pad (invalid: never in range)
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int j = wave + 12 * i;
    if (j < NYI) {
      lpx[i] = 4 * j + lp;
      lqy[i] = 0;
      loff[i] = (unsigned)((lpx[i] * p.dy.ld + co0 + lc4 * 4) * 4);
    } else {
      const int q = 4 * (j - NYI) + lp;
      const int qy = q / XW, xi = q - qy * XW;
      lpx[i] = xi - p.pad;
      lqy[i] = (qy < 3 && ci_ok) ? qy - p.pad : (1 << 28);
      loff[i] = (unsigned)((((qy - p.pad) * wx + xi - p.pad) * p.x.ld + ci0 + lc4 * 4) * 4);
    }
  }
  // LDS destinations of the three ops inside a ring slot (floats)
  const int ldst0 = op0_y ? wave * 256 : YSZ + (wave - NYI) * 256;
  const int ldst1 = YSZ + (wave + 12 - NYI) * 256, ldst2 = YSZ + (wave + 24 - NYI) * 256;
  int cur_n, cur_oy, cur_seg;
  {
    const WgChunk c = wg_chunk(p, c_begin < p.total_chunks ? c_begin : 0, P);
    cur_n = c.n; cur_oy = c.oy; cur_seg = c.ox0 / P;
  }
  // byte offsets of the cursor's row start (32-bit: the DMA path requires tensors below 4 GB), advanced incrementally
  unsigned yrow = (unsigned)((df_img_base(p.dy, cur_n) + (int64_t)cur_oy * wy * p.dy.ld) * 4);
  unsigned xrow = (unsigned)((df_img_base(p.x, cur_n) + (int64_t)cur_oy * STRIDE * wx * p.x.ld) * 4);
  const unsigned yrow_step = (unsigned)(wy * p.dy.ld * 4), xrow_step = (unsigned)(STRIDE * wx * p.x.ld * 4);
  const unsigned yseg_step = (unsigned)(P * p.dy.ld * 4), xseg_step = (unsigned)(P * STRIDE * p.x.ld * 4);
  auto issue = [&](int buf) {   // loads the cursor's chunk into ring slot `buf`, then advances the cursor
    float* slot = lds + buf * STG;
    const int ox0 = cur_seg * P;
    const unsigned ybase = yrow + (unsigned)cur_seg * yseg_step, xbase = xrow + (unsigned)cur_seg * xseg_step;
    const int iy0 = cur_oy * STRIDE, ix0 = ox0 * STRIDE;
    if (op0_y) {
      const bool ok = ox0 + lpx[0] < wy;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(yr, (lds_ptr_t)(slot + ldst0), 16, ok ? ybase + loff[0] : DMA_BAD, 0, 0, 0);
    } else {
      const bool ok = (unsigned)(iy0 + lqy[0]) < (unsigned)hx && (unsigned)(ix0 + lpx[0]) < (unsigned)wx;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr_t)(slot + ldst0), 16, ok ? xbase + loff[0] : DMA_BAD, 0, 0, 0);
    }
    {
      const bool ok = (unsigned)(iy0 + lqy[1]) < (unsigned)hx && (unsigned)(ix0 + lpx[1]) < (unsigned)wx;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr_t)(slot + ldst1), 16, ok ? xbase + loff[1] : DMA_BAD, 0, 0, 0);
    }
    if (op2_on) {
      const bool ok = (unsigned)(iy0 + lqy[2]) < (unsigned)hx && (unsigned)(ix0 + lpx[2]) < (unsigned)wx;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr_t)(slot + ldst2), 16, ok ? xbase + loff[2] : DMA_BAD, 0, 0, 0);
    }
    if (++cur_seg == p.chunks_per_row) {
      cur_seg = 0;
      yrow += yrow_step;
      xrow += xrow_step;
      if (++cur_oy == p.dy.h) {   // next image: its base need not follow the previous one
        cur_oy = 0;
        ++cur_n;
        yrow = (unsigned)(df_img_base(p.dy, cur_n) * 4);
        xrow = (unsigned)(df_img_base(p.x, cur_n) * 4);
      }
    }
  };

  static_assert(D >= 2 && D <= 4, "ring depth");
  static_assert(NOPS <= 36 && P % 8 == 0, "at most three DMA ops per wave and stage");
#pragma unroll
  for (int d = 0; d < D - 1; ++d)
    if (d < nst) issue(d);
  for (int i = 0; i < nst; ++i) {
    // this wave's DMA share of stage i has landed (D > 2: the ops of the D - 2 stages after it may still be in flight) ...
    if constexpr (D == 2) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      switch (min(D - 2, nst - 1 - i) * my_ops) {
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      }
    }
    __syncthreads();   // ... and everyone's; every wave has also finished reading ring slot (i - 1) % D
    if (i + D - 1 < nst) issue((i + D - 1) % D);
    const float* dyb = lds + (i % D) * STG;
    const float* xbuf = dyb + YSZ;
    if (wave_active && BF) {
      // bf16 operands.  The three horizontal taps of a lane read overlapping pixel runs: ten consecutive pixels are read
      // once (stride 1) and packed three times with a shift, instead of 3 x 8 LDS reads
#pragma unroll
      for (int ks = 0; ks < P / 16; ++ks) {
        float av[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) av[k] = dyb[(16 * ks + 8 * kh + k) * LC + wco * 32 + li];
        const bf16x8_t a8 = pack_bf16(av);
        if constexpr (STRIDE == 1) {
          float xv[10];
#pragma unroll
          for (int k = 0; k < 10; ++k) xv[k] = xbuf[(ky * XW + 16 * ks + 8 * kh + k) * LC + wci * 32 + li];
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            float bv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) bv[k] = xv[k + kx];
            acc[kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, pack_bf16(bv), acc[kx], 0, 0, 0);
          }
        } else {
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            float bv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) bv[k] = xbuf[(ky * XW + (16 * ks + 8 * kh + k) * STRIDE + kx) * LC + wci * 32 + li];
            acc[kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, pack_bf16(bv), acc[kx], 0, 0, 0);
          }
        }
      }
    } else if (wave_active) {
#pragma unroll 4
      for (int ks = 0; ks < P / 2; ++ks) {
        const int px = 2 * ks + kh;
        const float a = dyb[px * LC + wco * 32 + li];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const float b = xbuf[(ky * XW + px * STRIDE + kx) * LC + wci * 32 + li];
          acc[kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[kx], 0, 0, 0);
        }
      }
    }
    if (do_bias && tid < 512) {   // column tid & 63, pixel group tid >> 6 (8 groups of 4 pixels; combined after the loop)
#pragma unroll
      for (int j = 0; j < P / 8; ++j) bsum += dyb[((tid >> 6) * (P / 8) + j) * LC + (tid & 63)];
    }
  }
  if (do_bias) {
    __syncthreads();
    if (tid < 512) lds[tid] = bsum;
    __syncthreads();
    if (tid < LC) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) t += lds[64 * w + tid];
      p.bias_ws[(int64_t)split * p.N + co0 + tid] = t;
    }
  }
  if (wave_active) {
    float* o = p.ws + (int64_t)split * p.N * 9 * p.K;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = co0 + wco * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
        const int ci = ci0 + wci * 32 + li;
        o[((int64_t)co * 9 + ky * 3 + kx) * p.K + ci] = acc[kx][e];
      }
  }
#endif
}

// ---- 3x3 stride-1 weight gradient of the bf16-STORAGE training mode: x and dy are bfloat16 in memory -------------------
// The 12-wave ring kernel above moves fp32 tiles through LDS and rounds the fragments after k-strided 32-bit reads (18 reads
// + 16 conversions per 3 MFMAs: 24 % matrix-pipe time, profiles/r02_pmc_conv_bf16.txt).  Here
//   * both tensors arrive as bf16 by LDS-DMA: half the bytes per stage (17 KB instead of 35), so the ring is FOUR stages
//     deep at the same two workgroups per CU -- three stages of prefetch against the ~2 us loaded round trip that bounded
//     the two-deep ring;
//   * the k index of this GEMM is the PIXEL, the slow index of both NHWC tiles: the fragments are read with gfx950's
//     transposing LDS read (ds_read_b64_tr_b16: a 16-lane group reads a [4 pixels][16 channels] block and every lane gets
//     its channel's 4 consecutive pixels) -- 2 reads per 8-deep operand, no conversion, no packing: 8 LDS instructions per
//     3 MFMAs instead of 18 + 16 VALU;
//   * the LDS image is built for that read: each tile is split in two 32-channel halves with 64-byte pixel rows,
//     [half][pixel][32 ch], so the 32 lanes of a service group (rows P..P+3 x 64 B) cover all 64 banks exactly once,
//     whatever the tap shift.  The DMA builds it for free: slot s of an op takes the 16-byte source chunk (pixel, 8 channels)
//     the image wants there (lane-linear destination, per-lane source address).
// Wave = (32 co x 32 ci quadrant) x kernel row ky as in the ring kernel; same accumulator layout and split-K partials.
template <int D>
__global__ __launch_bounds__(768) void wgrad3_tr_kernel(WgradParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int P = 32, XW = P + 2, LC = 64;
  constexpr int YB = 2 * P * 64;                        // dY tile bytes: [2 halves][32 px][64 B]          = 4096
  constexpr int XH = 3 * XW * 64;                       // one X half: [3 rows x 34 px][64 B]              = 6528
  constexpr int NYO = YB / 1024, NXO = (2 * XH + 1023) / 1024;   // 1-KB DMA ops: 4 + 13
  constexpr int XB = NXO * 1024, STG = YB + XB;         // stage bytes (17408)
  constexpr int NOPS = NYO + NXO;                       // 17 ops per stage over 12 waves: op j = wave (+ 12)
  static_assert(NYO <= 12 && NOPS > 12 && NOPS <= 24, "op 0: dY or X, op 1: X");
  extern __shared__ __attribute__((aligned(16))) char ldsb[];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  typedef __attribute__((address_space(3))) s16x4* lds_s4_t;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kh = lane >> 5;
  const int quad = wave & 3, ky = wave >> 2;
  const int wci = quad & 1, wco = quad >> 1;
  int bx = blockIdx.x, by = blockIdx.y, split = blockIdx.z;
  if (p.xcd_map) {   // all (ci, co) tiles of a split on one XCD: they read the same x / dy tiles (see wgrad3_ring_kernel)
    const int nt = gridDim.x * gridDim.y;
    const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int lg = df_xcd_swizzle(lin, nt * gridDim.z);
    const int tile = lg % nt;
    split = lg / nt;
    bx = tile % gridDim.x;
    by = tile / gridDim.x;
  }
  const int ci0 = bx * LC, co0 = by * LC;
  const bool do_bias = p.bias_ws && bx == 0;
  float bsum = 0.f;

  f32x16 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(p.x.ptr, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.dy.ptr, 0, p.dy_bytes, 0x00020000);
  const int wy = p.dy.w, hx = p.x.h, wx = p.x.w;
  const int c_begin = split * p.chunks_per_split;
  const int c_end = min(c_begin + p.chunks_per_split, p.total_chunks);
  const int nst = max(c_end - c_begin, 0);

  // DMA ops: lane-constant source parts (bytes), wave-uniform cursor for the rest (no divisions in the loop)
  const bool op0_y = wave < NYO, op1_on = wave + 12 < NOPS;
  unsigned loff[2];
  int lpx[2], lqy[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int j = wave + 12 * i;
    if (j < NYO) {
      const int s = j * 64 + lane;                       // 16-byte slot of the dY image
      const int h = s >> 7, px = (s & 127) >> 2, q = s & 3;
      lpx[i] = px;
      lqy[i] = 0;
      loff[i] = (unsigned)((px * p.dy.ld + co0 + 32 * h + 8 * q) * 2);
    } else {
      const int s = (j - NYO) * 64 + lane;               // slot of the X image: [half][row qy][xi][4 slots]
      const int h = s / (3 * XW * 4), r = (s - h * 3 * XW * 4) >> 2, q = s & 3;
      const int qy = r / XW, xi = r - qy * XW;
      const bool on = h < 2 && j < NOPS && (ci0 + 32 * h + 8 * q) < p.K;
      lpx[i] = xi - 1;
      lqy[i] = on ? qy - 1 : (1 << 28);
      loff[i] = (unsigned)((((qy - 1) * wx + xi - 1) * p.x.ld + ci0 + 32 * h + 8 * q) * 2);
    }
  }
  const int ldst0 = op0_y ? wave * 1024 : YB + (wave - NYO) * 1024;
  const int ldst1 = YB + (wave + 12 - NYO) * 1024;
  int cur_n, cur_oy, cur_seg;      // column-major chunk order (see wgrad3_h2p_kernel): chunk -> (n, segment, oy), oy fastest
  {
    const int ch = c_begin < p.total_chunks ? c_begin : 0;
    const int per_img = p.chunks_per_row * p.dy.h;
    cur_n = ch / per_img;
    const int rem = ch - cur_n * per_img;
    cur_seg = rem / p.dy.h;
    cur_oy = rem - cur_seg * p.dy.h;
  }
  unsigned yrow = (unsigned)((df_img_base(p.dy, cur_n) + (int64_t)cur_oy * wy * p.dy.ld) * 2);
  unsigned xrow = (unsigned)((df_img_base(p.x, cur_n) + (int64_t)cur_oy * wx * p.x.ld) * 2);
  const unsigned yrow_step = (unsigned)(wy * p.dy.ld * 2), xrow_step = (unsigned)(wx * p.x.ld * 2);
  const unsigned yseg_step = (unsigned)(P * p.dy.ld * 2), xseg_step = (unsigned)(P * p.x.ld * 2);
  auto issue = [&](int buf) {   // loads the cursor's chunk into ring slot `buf`, then advances the cursor
    char* slot = ldsb + buf * STG;
    const int ox0 = cur_seg * P;
    const unsigned ybase = yrow + (unsigned)cur_seg * yseg_step, xbase = xrow + (unsigned)cur_seg * xseg_step;
    if (op0_y) {
      const bool ok = ox0 + lpx[0] < wy;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(yr, (lds_ptr_t)(slot + ldst0), 16, ok ? ybase + loff[0] : DMA_BAD, 0, 0, 0);
    } else {
      const bool ok = (unsigned)(cur_oy + lqy[0]) < (unsigned)hx && (unsigned)(ox0 + lpx[0]) < (unsigned)wx;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr_t)(slot + ldst0), 16, ok ? xbase + loff[0] : DMA_BAD, 0, 0, 0);
    }
    if (op1_on) {
      const bool ok = (unsigned)(cur_oy + lqy[1]) < (unsigned)hx && (unsigned)(ox0 + lpx[1]) < (unsigned)wx;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr_t)(slot + ldst1), 16, ok ? xbase + loff[1] : DMA_BAD, 0, 0, 0);
    }
    yrow += yrow_step;
    xrow += xrow_step;
    if (++cur_oy == p.dy.h) {     // next column segment of the image, or the next image
      cur_oy = 0;
      if (++cur_seg == p.chunks_per_row) {
        cur_seg = 0;
        ++cur_n;
      }
      yrow = (unsigned)(df_img_base(p.dy, cur_n) * 2);
      xrow = (unsigned)(df_img_base(p.x, cur_n) * 2);
    }
  };

  // transposing reads: lane (group g = lane >> 4: channel block cb = g & 1, k half kh = g >> 1; i = lane & 15) supplies the
  // address of pixel row (i >> 2), 8-byte column chunk (i & 3) of its 16-channel block and receives channel (cb * 16 + i)'s
  // 4 consecutive pixels.  Issued as inline assembly: through the builtin the compiler treats every read as possibly aliasing
  // the LDS-DMA in flight and puts s_waitcnt vmcnt(0) in front of the first one -- the whole prefetched ring drained per stage.
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)ldsb;
  const int tr_lane = ((lane & 15) >> 2) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;    // bytes inside a half
  const unsigned a_base = lds0 + wco * (P * 64) + (8 * kh) * 64 + tr_lane;                   // dY half wco, pixel 8 kh
  const unsigned b_base = lds0 + YB + wci * XH + (ky * XW + 8 * kh) * 64 + tr_lane;          // X half wci, row ky, pixel 8 kh
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  auto op8 = [](u32x2_t lo, u32x2_t hi) -> bf16x8_t {
    u32x4_t v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = hi[0]; v[3] = hi[1];
    return __builtin_bit_cast(bf16x8_t, v);
  };
  static_assert(D >= 2 && D <= 4, "ring depth");
  const int my_ops = op1_on ? 2 : 1;
#pragma unroll
  for (int d = 0; d < D - 1; ++d)
    if (d < nst) issue(d);
  for (int i = 0; i < nst; ++i) {
    // this wave's DMA share of stage i has landed (the ops of up to D - 2 later stages may still be in flight) ...
    switch (min(D - 2, nst - 1 - i) * my_ops) {
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
    // ... and everyone's; every wave has also finished reading ring slot (i - 1) % D.  A RAW barrier: __syncthreads() carries a
    // fence for which the compiler drains vmcnt to 0 -- i.e. waits for the whole prefetched ring -- in front of it
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (i + D - 1 < nst) issue((i + D - 1) % D);
    const char* st = ldsb + (i % D) * STG;
    const unsigned so = (unsigned)((i % D) * STG);
#pragma unroll
    for (int ks = 0; ks < P / 16; ++ks) {
      // the 16-pixel step's operands: dY pixels +0..3 / +4..7, and the same for the three horizontal taps of x (one pixel
      // = 64 bytes further per tap)
      u32x2_t al, ah, b0l, b0h, b1l, b1h, b2l, b2h;
      const unsigned aa = a_base + so + ks * 16 * 64, ba = b_base + so + ks * 16 * 64;
      asm volatile(
          "ds_read_b64_tr_b16 %0, %8\n\t"
          "ds_read_b64_tr_b16 %1, %8 offset:256\n\t"
          "ds_read_b64_tr_b16 %2, %9\n\t"
          "ds_read_b64_tr_b16 %3, %9 offset:256\n\t"
          "ds_read_b64_tr_b16 %4, %9 offset:64\n\t"
          "ds_read_b64_tr_b16 %5, %9 offset:320\n\t"
          "ds_read_b64_tr_b16 %6, %9 offset:128\n\t"
          "ds_read_b64_tr_b16 %7, %9 offset:384\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "=&v"(al), "=&v"(ah), "=&v"(b0l), "=&v"(b0h), "=&v"(b1l), "=&v"(b1h), "=&v"(b2l), "=&v"(b2h)
          : "v"(aa), "v"(ba)
          : "memory");
      const bf16x8_t a8 = op8(al, ah);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, op8(b0l, b0h), acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, op8(b1l, b1h), acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, op8(b2l, b2h), acc[2], 0, 0, 0);
    }
    if (do_bias && tid < 512) {   // column tid & 63, pixel group tid >> 6 (8 groups of 4 pixels; combined after the loop)
      const unsigned short* dy16 = reinterpret_cast<const unsigned short*>(st);
      const int c = tid & 63;
#pragma unroll
      for (int j = 0; j < P / 8; ++j)
        bsum += __builtin_bit_cast(float, (unsigned)dy16[(c >> 5) * (P * 32) + ((tid >> 6) * (P / 8) + j) * 32 + (c & 31)] << 16);
    }
  }
  if (do_bias) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(ldsb);
    if (tid < 512) red[tid] = bsum;
    __syncthreads();
    if (tid < LC) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) t += red[64 * w + tid];
      p.bias_ws[(int64_t)split * p.N + co0 + tid] = t;
    }
  }
  if ((ci0 + wci * 32) < p.K) {
    float* o = p.ws + (int64_t)split * p.N * 9 * p.K;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = co0 + wco * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
        const int ci = ci0 + wci * 32 + li;
        o[((int64_t)co * 9 + ky * 3 + kx) * p.K + ci] = acc[kx][e];
      }
  }
#endif
}

// ---- fp32-ACCURATE 3x3 stride-1 weight gradient on the bf16 matrix pipe (bf16x3, df_conv2d_wgrad_x3) ---------------------
// The weight-gradient twin of conv_halo_x3_kernel: x and dy are fp32 in memory; every staged element is split once into three
// bf16 planes (hi + mid + lo == the fp32 value) on its way global -> registers -> LDS, the planes are laid out as the
// transposing-read image of wgrad3_tr_kernel ([plane][half][pixel][32 ch], 64-byte pixel rows), and each 8-deep operand pair
// is multiplied as six exact bf16 products -- 36 MFMAs per wave and 32-pixel stage, dW to fp32 rounding at 16 / 6 of the fp32
// MFMA rate.  12 waves = (32 co x 32 ci quadrant) x kernel row; two LDS stages of 3 x 17 KB (one workgroup per CU); the next
// stage's elements are fetched into registers while the current stage is multiplied.
// NP = 2: two fp16 planes per operand with per-tensor power-of-two scales (see conv_halo_x3_kernel): three MFMAs per operand pair
// instead of six, two accumulators per tap (hi.hi | cross terms x 2048), 2 / 3 of the LDS bytes.
template <int NP>
__global__ __launch_bounds__(768) void wgrad3_x3_kernel(WgradParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int P = 32, XW = P + 2, LC = 64;
  constexpr int YB = 2 * P * 64;                        // dY image bytes of one plane: [2 halves][32 px][64 B]
  constexpr int XH = 3 * XW * 64;                       // one X half: [3 rows x 34 px][64 B]
  constexpr int PLB = YB + 2 * XH;                      // one plane of a stage (17152 B)
  constexpr int STG = NP * PLB;                         // stage bytes (51456 / 34304)
  static_assert(NP == 2 || NP == 3, "planes");
  float sx = 1.f, sdy = 1.f;
  if constexpr (NP == 2) { sx = df_h2_scale(*p.amax_x); sdy = df_h2_scale(*p.amax_dy); }
  constexpr int NYS = YB / 16, NXS = 2 * XH / 16;       // 16-byte slots: 256 + 816
  constexpr int NIT = (NYS + NXS + 767) / 768;          // items per thread (2)
  extern __shared__ __attribute__((aligned(16))) char ldsb[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kh = lane >> 5;
  const int quad = wave & 3, ky = wave >> 2;
  const int wci = quad & 1, wco = quad >> 1;
  int bx = blockIdx.x, by = blockIdx.y, split = blockIdx.z;
  if (p.xcd_map) {   // all (ci, co) tiles of a split on one XCD: they read the same x / dy tiles (see wgrad3_ring_kernel)
    const int nt = gridDim.x * gridDim.y;
    const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int lg = df_xcd_swizzle(lin, nt * gridDim.z);
    const int tile = lg % nt;
    split = lg / nt;
    bx = tile % gridDim.x;
    by = tile / gridDim.x;
  }
  const int ci0 = bx * LC, co0 = by * LC;
  const bool do_bias = p.bias_ws && bx == 0;
  float bsum = 0.f;

  f32x16 acc[3];
  f32x16 acc1[NP == 2 ? 3 : 1];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      acc[t][e] = 0.f;
      if constexpr (NP == 2) acc1[t][e] = 0.f;
    }

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(p.x.ptr, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.dy.ptr, 0, p.dy_bytes, 0x00020000);
  const int wy = p.dy.w, hx = p.x.h, wx = p.x.w;
  const int c_begin = split * p.chunks_per_split;
  const int c_end = min(c_begin + p.chunks_per_split, p.total_chunks);
  const int nst = max(c_end - c_begin, 0);

  // staging items: item e of a thread = 16-byte slot (tid + 768 e) of a plane image: slots 0 .. 255 = dY ([half][px][4]), the
  // rest = X ([half][row][xi][4]); the source is 8 consecutive fp32 channels (two 16-byte loads)
  unsigned loff[NIT];     // lane-constant source byte offset relative to the stage's row / segment base
  int lpx[NIT], lqy[NIT]; // dY: pixel (lqy = -100); X: column offset xi - 1, row offset qy - 1 (off: never in range)
  int ldst[NIT];          // destination byte inside a plane
#pragma unroll
  for (int e = 0; e < NIT; ++e) {
    const int t = tid + 768 * e;
    if (t < NYS) {
      const int h = t >> 7, px = (t & 127) >> 2, q = t & 3;
      lpx[e] = px;
      lqy[e] = -100;
      loff[e] = (unsigned)((px * p.dy.ld + co0 + 32 * h + 8 * q) * 4);
      ldst[e] = 16 * t;
    } else {
      const int sx = t - NYS;
      const int h = sx / (3 * XW * 4), r = (sx - h * 3 * XW * 4) >> 2, q = sx & 3;
      const int qy = r / XW, xi = r - qy * XW;
      const bool on = sx < NXS && (ci0 + 32 * h + 8 * q) < p.K;
      lpx[e] = xi - 1;
      lqy[e] = on ? qy - 1 : (1 << 28);
      loff[e] = (unsigned)((((qy - 1) * wx + xi - 1) * p.x.ld + ci0 + 32 * h + 8 * q) * 4);
      ldst[e] = sx < NXS ? YB + 16 * sx : -1;
    }
  }
  int cur_n, cur_oy, cur_seg;
  {
    const WgChunk c = wg_chunk(p, c_begin < p.total_chunks ? c_begin : 0, P);
    cur_n = c.n; cur_oy = c.oy; cur_seg = c.ox0 / P;
  }
  unsigned yrow = (unsigned)((df_img_base(p.dy, cur_n) + (int64_t)cur_oy * wy * p.dy.ld) * 4);
  unsigned xrow = (unsigned)((df_img_base(p.x, cur_n) + (int64_t)cur_oy * wx * p.x.ld) * 4);
  const unsigned yrow_step = (unsigned)(wy * p.dy.ld * 4), xrow_step = (unsigned)(wx * p.x.ld * 4);
  const unsigned yseg_step = (unsigned)(P * p.dy.ld * 4), xseg_step = (unsigned)(P * p.x.ld * 4);
  f32x4 ra[NIT][2];
  auto fetch = [&]() {          // the cursor's chunk -> registers, then advance the cursor
    const int ox0 = cur_seg * P;
    const unsigned ybase = yrow + (unsigned)cur_seg * yseg_step, xbase = xrow + (unsigned)cur_seg * xseg_step;
#pragma unroll
    for (int e = 0; e < NIT; ++e) {
      unsigned v;
      if (lqy[e] == -100) {
        v = (ox0 + lpx[e] < wy) ? ybase + loff[e] : DMA_BAD;
        ra[e][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(yr, v, 0, 0));
        ra[e][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(yr, v + 16, 0, 0));
      } else {
        const bool ok = (unsigned)(cur_oy + lqy[e]) < (unsigned)hx && (unsigned)(ox0 + lpx[e]) < (unsigned)wx;
        v = ok ? xbase + loff[e] : DMA_BAD;
        ra[e][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, v, 0, 0));
        ra[e][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, v + 16, 0, 0));
      }
    }
    if (++cur_seg == p.chunks_per_row) {
      cur_seg = 0;
      yrow += yrow_step;
      xrow += xrow_step;
      if (++cur_oy == p.dy.h) {
        cur_oy = 0;
        ++cur_n;
        yrow = (unsigned)(df_img_base(p.dy, cur_n) * 4);
        xrow = (unsigned)(df_img_base(p.x, cur_n) * 4);
      }
    }
  };
  auto stash = [&](int buf) {   // registers -> hi / mid / lo planes -> LDS
    char* st = ldsb + buf * STG;
#pragma unroll
    for (int e = 0; e < NIT; ++e) {
      if (ldst[e] >= 0) {
        float v[8], r[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] = ra[e][0][k]; v[4 + k] = ra[e][1][k]; }
        if constexpr (NP == 2) {
          f16x8_t h, l;
          df_h2_split(v, lqy[e] == -100 ? sdy : sx, h, l);
          char* d = st + ldst[e];
          *reinterpret_cast<f16x8_t*>(d) = h;
          *reinterpret_cast<f16x8_t*>(d + PLB) = l;
          continue;
        }
        bf16x8_t hi, mi, lo;
#pragma unroll
        for (int k = 0; k < 8; ++k) { hi[k] = (__bf16)v[k]; r[k] = v[k] - (float)hi[k]; }
#pragma unroll
        for (int k = 0; k < 8; ++k) { mi[k] = (__bf16)r[k]; r[k] = r[k] - (float)mi[k]; }
#pragma unroll
        for (int k = 0; k < 8; ++k) lo[k] = (__bf16)r[k];
        char* d = st + ldst[e];
        *reinterpret_cast<bf16x8_t*>(d) = hi;
        *reinterpret_cast<bf16x8_t*>(d + PLB) = mi;
        *reinterpret_cast<bf16x8_t*>(d + 2 * PLB) = lo;
      }
    }
  };

  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)ldsb;
  const int tr_lane = ((lane & 15) >> 2) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;    // (see wgrad3_tr_kernel)
  const unsigned a_base = lds0 + wco * (P * 64) + (8 * kh) * 64 + tr_lane;
  const unsigned b_base = lds0 + YB + wci * XH + (ky * XW + 8 * kh) * 64 + tr_lane;
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  auto op8 = [](u32x2_t lo, u32x2_t hi) -> bf16x8_t {
    u32x4_t v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = hi[0]; v[3] = hi[1];
    return __builtin_bit_cast(bf16x8_t, v);
  };

  if (nst > 0) {
    fetch();
    stash(0);
  }
  __syncthreads();
  for (int i = 0; i < nst; ++i) {
    if (i + 1 < nst) fetch();
    const unsigned so = (unsigned)((i & 1) * STG);
#pragma unroll
    for (int ks = 0; ks < P / 16; ++ks) {
      const unsigned aa = a_base + so + ks * 16 * 64, ba = b_base + so + ks * 16 * 64;
      if constexpr (NP == 2) {
        u32x2_t ahl, ahh, all_, alh;
        asm volatile(
            "ds_read_b64_tr_b16 %0, %4\n\t"
            "ds_read_b64_tr_b16 %1, %4 offset:256\n\t"
            "ds_read_b64_tr_b16 %2, %4 offset:17152\n\t"
            "ds_read_b64_tr_b16 %3, %4 offset:17408\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(ahl), "=&v"(ahh), "=&v"(all_), "=&v"(alh)
            : "v"(aa)
            : "memory");
        const f16x8_t ah = __builtin_bit_cast(f16x8_t, op8(ahl, ahh)), al = __builtin_bit_cast(f16x8_t, op8(all_, alh));
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          u32x2_t bhl, bhh, bll, blh;
          const unsigned bb = ba + kx * 64;
          asm volatile(
              "ds_read_b64_tr_b16 %0, %4\n\t"
              "ds_read_b64_tr_b16 %1, %4 offset:256\n\t"
              "ds_read_b64_tr_b16 %2, %4 offset:17152\n\t"
              "ds_read_b64_tr_b16 %3, %4 offset:17408\n\t"
              "s_waitcnt lgkmcnt(0)"
              : "=&v"(bhl), "=&v"(bhh), "=&v"(bll), "=&v"(blh)
              : "v"(bb)
              : "memory");
          const f16x8_t bh = __builtin_bit_cast(f16x8_t, op8(bhl, bhh)), bl = __builtin_bit_cast(f16x8_t, op8(bll, blh));
          acc1[kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc1[kx], 0, 0, 0);
          acc[kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[kx], 0, 0, 0);
          acc1[kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc1[kx], 0, 0, 0);
        }
        continue;
      }
      // dY operand: three planes x (pixels +0..3, +4..7); plane stride PLB = 17152 bytes (immediate offsets)
      u32x2_t ahl, ahh, aml, amh, all_, alh;
      asm volatile(
          "ds_read_b64_tr_b16 %0, %6\n\t"
          "ds_read_b64_tr_b16 %1, %6 offset:256\n\t"
          "ds_read_b64_tr_b16 %2, %6 offset:17152\n\t"
          "ds_read_b64_tr_b16 %3, %6 offset:17408\n\t"
          "ds_read_b64_tr_b16 %4, %6 offset:34304\n\t"
          "ds_read_b64_tr_b16 %5, %6 offset:34560\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "=&v"(ahl), "=&v"(ahh), "=&v"(aml), "=&v"(amh), "=&v"(all_), "=&v"(alh)
          : "v"(aa)
          : "memory");
      const bf16x8_t ah = op8(ahl, ahh), am = op8(aml, amh), al = op8(all_, alh);
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        u32x2_t bhl, bhh, bml, bmh, bll, blh;
        const unsigned bb = ba + kx * 64;
        asm volatile(
            "ds_read_b64_tr_b16 %0, %6\n\t"
            "ds_read_b64_tr_b16 %1, %6 offset:256\n\t"
            "ds_read_b64_tr_b16 %2, %6 offset:17152\n\t"
            "ds_read_b64_tr_b16 %3, %6 offset:17408\n\t"
            "ds_read_b64_tr_b16 %4, %6 offset:34304\n\t"
            "ds_read_b64_tr_b16 %5, %6 offset:34560\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(bhl), "=&v"(bhh), "=&v"(bml), "=&v"(bmh), "=&v"(bll), "=&v"(blh)
            : "v"(bb)
            : "memory");
        const bf16x8_t bh = op8(bhl, bhh), bm = op8(bml, bmh), bl = op8(bll, blh);
        f32x16 c = acc[kx];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);   // small terms first
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
        acc[kx] = c;
      }
    }
    if (do_bias && tid < 512) {   // column tid & 63, pixel group tid >> 6: the exact fp32 values are hi + mid + lo
      const char* stp = ldsb + (i & 1) * STG;
      const int c = tid & 63;
#pragma unroll
      for (int j = 0; j < P / 8; ++j) {
        const int el = (c >> 5) * (P * 32) + ((tid >> 6) * (P / 8) + j) * 32 + (c & 31);
        float v = 0.f;
        if constexpr (NP == 2) {
          v = (float)reinterpret_cast<const _Float16*>(stp)[el] + (float)reinterpret_cast<const _Float16*>(stp + PLB)[el] * H2_LO_INV;
        } else {
#pragma unroll
          for (int pl = 2; pl >= 0; --pl)
            v += __builtin_bit_cast(float, (unsigned)reinterpret_cast<const unsigned short*>(stp + pl * PLB)[el] << 16);
        }
        bsum += v;
      }
    }
    if (i + 1 < nst) stash((i + 1) & 1);    // ring slot (i + 1) & 1 was last read in stage i - 1, behind the previous barrier
    __syncthreads();
  }
  if (do_bias) {
    float* red = reinterpret_cast<float*>(ldsb);
    if (tid < 512) red[tid] = bsum;
    __syncthreads();
    if (tid < LC) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) t += red[64 * w + tid];
      p.bias_ws[(int64_t)split * p.N + co0 + tid] = NP == 2 ? t * (1.f / sdy) : t;
    }
  }
  if ((ci0 + wci * 32) < p.K) {
    float* o = p.ws + (int64_t)split * p.N * 9 * p.K;
    const float ix = 1.f / sx, iy = 1.f / sdy;      // (NP == 2) exact powers of two
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = co0 + wco * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
        const int ci = ci0 + wci * 32 + li;
        float v = acc[kx][e];
        if constexpr (NP == 2) v = (v + acc1[kx][e] * H2_LO_INV) * ix * iy;
        o[((int64_t)co * 9 + ky * 3 + kx) * p.K + ci] = v;
      }
  }
#endif
}

// ---- 1x1 weight gradient on the fp16 matrix pipe, fp32 tensors split in flight (round 5; df_conv2d_wgrad1_h2) ----------------
// dW[co, ci] = sum_p dy[p, co] x[p, ci] is a row GEMM whose K dimension is the pixel count: every operand byte is read once and the
// fp32 MFMA (wgrad1x1_kernel: 32x32x2, 64 cycles each) cannot keep up with the memory system -- 82-88 TFLOP/s = 1.4-2.6 TB/s of
// operands on the decoder's skip / latent layers, 55-60 with the 64 x 64 tile of wgrad_kernel<1,1,32> (profiles/r05_layer_table.txt).
// Here every staged element is split ONCE into two scaled fp16 planes (df_h2_split, the scales from bounds of max |x| / max |dy| as in
// wgrad3_x3_kernel<2>) on its way global -> registers -> LDS, the planes take the transposing-read image of wgrad3_tr_kernel
// ([plane][32-channel half][pixel][64 B]) and a product is three v_mfma_f32_32x32x16_f16 -- 5.3 x the fp32 MFMA rate, which leaves
// the kernel to the memory system.  COT x CIT tile (128 or 64 each), 4 waves = 2 x 2 wave tiles, 32 pixels per stage, two LDS
// stages, the next stage's elements in registers while the current one is multiplied; two workgroups per CU.
// BF (bf16 MFMA mode, Trainer(dtype="bf16")): ONE bf16 plane per operand (the fp32 element rounded once), one MFMA per product, no scales.
template <int COT, int CIT, bool BF = false>
__global__ __launch_bounds__(256, 2) void wgrad1_h2_kernel(WgradParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int P = 32;
  constexpr int HY = COT / 32, HX = CIT / 32;          // 32-channel halves of the two operands
  constexpr int HB = P * 64;                           // bytes of one half of one plane: [32 px][64 B]
  constexpr int YB = HY * HB, XB = HX * HB;
  constexpr int PLB = YB + XB, STG = (BF ? 1 : 2) * PLB;   // plane, stage (hi | lo; BF: one plane)
  constexpr int NYS = YB / 16, NXS = XB / 16;          // 16-byte slots = 8 channels of one pixel of one plane
  constexpr int NIT = (NYS + NXS) / 256, NIY = NYS / 256;
  constexpr int TCO = COT / 64, TCI = CIT / 64;        // 32-wide tiles per wave and operand
  static_assert(NYS % 256 == 0 && NXS % 256 == 0 && (COT == 64 || COT == 128) && (CIT == 64 || CIT == 128), "tile");
  float sx = 1.f, sdy = 1.f;
  if constexpr (!BF) { sx = df_h2_scale(*p.amax_x); sdy = df_h2_scale(*p.amax_dy); }
  extern __shared__ __attribute__((aligned(16))) char ldsb[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kh = lane >> 5;
  const int wco = wave & 1, wci = wave >> 1;
  int bx = blockIdx.x, by = blockIdx.y, split = blockIdx.z;
  if (p.xcd_map) {   // all (ci, co) tiles of a split on one XCD: they read the same x / dy rows (see wgrad3_ring_kernel)
    const int nt = gridDim.x * gridDim.y;
    const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int lg = df_xcd_swizzle(lin, nt * gridDim.z);
    const int tile = lg % nt;
    split = lg / nt;
    bx = tile % gridDim.x;
    by = tile / gridDim.x;
  }
  const int ci0 = bx * CIT, co0 = by * COT;
  const bool do_bias = p.bias_ws && bx == 0;
  float bsum = 0.f;

  f32x16 acc[TCO][TCI], acc1[TCO][TCI];
#pragma unroll
  for (int i = 0; i < TCO; ++i)
#pragma unroll
    for (int j = 0; j < TCI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) { acc[i][j][e] = 0.f; acc1[i][j][e] = 0.f; }

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(p.x.ptr, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.dy.ptr, 0, p.dy_bytes, 0x00020000);
  const int wy = p.dy.w;
  const int c_begin = split * p.chunks_per_split;
  const int c_end = min(c_begin + p.chunks_per_split, p.total_chunks);
  const int nst = max(c_end - c_begin, 0);

  // staging items: item e of a thread = slot tid + 256 e of a plane image -- items < NIY are dY ([half][px][4 slots]), the rest X;
  // the source of a slot is 8 consecutive fp32 channels (two 16-byte loads)
  unsigned loff[NIT];
  int lpx[NIT];
#pragma unroll
  for (int e = 0; e < NIT; ++e) {
    const int t = (tid + 256 * e) - (e < NIY ? 0 : NYS);
    const int h = t >> 7, px = (t & 127) >> 2, q = t & 3;
    if (e < NIY) {
      lpx[e] = px;
      loff[e] = (unsigned)((px * p.dy.ld + co0 + 32 * h + 8 * q) * 4);
    } else {
      lpx[e] = (ci0 + 32 * h + 8 * q) < p.K ? px : (1 << 28);       // (channel groups past K: never in range -> zeros)
      loff[e] = (unsigned)((px * p.x.ld + ci0 + 32 * h + 8 * q) * 4);
    }
  }
  int cur_n, cur_oy, cur_seg;
  {
    const WgChunk c = wg_chunk(p, c_begin < p.total_chunks ? c_begin : 0, P);
    cur_n = c.n; cur_oy = c.oy; cur_seg = c.ox0 / P;
  }
  unsigned yrow = (unsigned)((df_img_base(p.dy, cur_n) + (int64_t)cur_oy * wy * p.dy.ld) * 4);
  unsigned xrow = (unsigned)((df_img_base(p.x, cur_n) + (int64_t)cur_oy * wy * p.x.ld) * 4);
  const unsigned yrow_step = (unsigned)(wy * p.dy.ld * 4), xrow_step = (unsigned)(wy * p.x.ld * 4);
  const unsigned yseg_step = (unsigned)(P * p.dy.ld * 4), xseg_step = (unsigned)(P * p.x.ld * 4);
  f32x4 ra[NIT][2];
  auto fetch = [&]() {          // the cursor's chunk -> registers, then advance the cursor
    const int ox0 = cur_seg * P;
    const unsigned ybase = yrow + (unsigned)cur_seg * yseg_step, xbase = xrow + (unsigned)cur_seg * xseg_step;
#pragma unroll
    for (int e = 0; e < NIT; ++e) {
      const bool ok = ox0 + lpx[e] < wy;
      if (e < NIY) {
        const unsigned v = ok ? ybase + loff[e] : DMA_BAD;
        ra[e][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(yr, v, 0, 0));
        ra[e][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(yr, v + 16, 0, 0));
      } else {
        const unsigned v = ok ? xbase + loff[e] : DMA_BAD;
        ra[e][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, v, 0, 0));
        ra[e][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, v + 16, 0, 0));
      }
    }
    if (++cur_seg == p.chunks_per_row) {
      cur_seg = 0;
      yrow += yrow_step;
      xrow += xrow_step;
      if (++cur_oy == p.dy.h) {
        cur_oy = 0;
        ++cur_n;
        yrow = (unsigned)(df_img_base(p.dy, cur_n) * 4);
        xrow = (unsigned)(df_img_base(p.x, cur_n) * 4);
      }
    }
  };
  auto stash = [&](int buf) {   // registers -> (hi, lo) fp16 planes -> LDS
    char* st = ldsb + buf * STG;
#pragma unroll
    for (int e = 0; e < NIT; ++e) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 4; ++k) { v[k] = ra[e][0][k]; v[4 + k] = ra[e][1][k]; }
      char* d = st + 16 * (tid + 256 * e);            // (dY slots first, X slots behind them: YB = 16 NYS)
      if constexpr (BF) {
        *reinterpret_cast<bf16x8_t*>(d) = pack_bf16(v);
      } else {
        f16x8_t h, l;
        df_h2_split(v, e < NIY ? sdy : sx, h, l);
        *reinterpret_cast<f16x8_t*>(d) = h;
        *reinterpret_cast<f16x8_t*>(d + PLB) = l;
      }
    }
  };

  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)ldsb;
  const int tr_lane = ((lane & 15) >> 2) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;    // (see wgrad3_tr_kernel)
  const unsigned a_base = lds0 + wco * TCO * HB + (8 * kh) * 64 + tr_lane;
  const unsigned b_base = lds0 + YB + wci * TCI * HB + (8 * kh) * 64 + tr_lane;
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  auto op8 = [](u32x2_t lo, u32x2_t hi) -> f16x8_t {
    u32x4_t v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = hi[0]; v[3] = hi[1];
    return __builtin_bit_cast(f16x8_t, v);
  };
  // one 32 x 16 operand tile = (hi, lo) x (pixels +0..3, +4..7): four transposing reads (BF: the one plane, two reads)
  auto rd4 = [](unsigned hi_addr, unsigned lo_addr, u32x2_t (&r)[4]) {
    if constexpr (BF) {
      asm volatile(
          "ds_read_b64_tr_b16 %0, %2\n\t"
          "ds_read_b64_tr_b16 %1, %2 offset:256"
          : "=&v"(r[0]), "=&v"(r[1])
          : "v"(hi_addr)
          : "memory");
    } else {
      asm volatile(
          "ds_read_b64_tr_b16 %0, %4\n\t"
          "ds_read_b64_tr_b16 %1, %4 offset:256\n\t"
          "ds_read_b64_tr_b16 %2, %5\n\t"
          "ds_read_b64_tr_b16 %3, %5 offset:256"
          : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3])
          : "v"(hi_addr), "v"(lo_addr)
          : "memory");
    }
  };
  // after the stage's s_waitcnt: ties a tile's registers to this point of the (volatile, hence ordered) asm sequence, so that no
  // product can be scheduled in front of the wait that covers its operands
  auto pin = [](u32x2_t (&r)[4]) {
    if constexpr (BF) asm volatile("" : "+v"(r[0]), "+v"(r[1]) :: "memory");
    else asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) :: "memory");
  };

  if (nst > 0) {
    fetch();
    stash(0);
  }
  __syncthreads();
  for (int i = 0; i < nst; ++i) {
    if (i + 1 < nst) fetch();
    const unsigned so = (unsigned)((i & 1) * STG);
#pragma unroll
    for (int ks = 0; ks < P / 16; ++ks) {
      u32x2_t fa[TCO][4], fb[TCI][4];
#pragma unroll
      for (int t = 0; t < TCO; ++t) rd4(a_base + so + ks * 1024 + t * HB, a_base + so + ks * 1024 + t * HB + PLB, fa[t]);
#pragma unroll
      for (int t = 0; t < TCI; ++t) rd4(b_base + so + ks * 1024 + t * HB, b_base + so + ks * 1024 + t * HB + PLB, fb[t]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int t = 0; t < TCO; ++t) pin(fa[t]);
#pragma unroll
      for (int t = 0; t < TCI; ++t) pin(fb[t]);
#pragma unroll
      for (int ti = 0; ti < TCO; ++ti) {
#pragma unroll
        for (int tj = 0; tj < TCI; ++tj) {
          if constexpr (BF) {
            acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, op8(fa[ti][0], fa[ti][1])),
                                                                  __builtin_bit_cast(bf16x8_t, op8(fb[tj][0], fb[tj][1])), acc[ti][tj], 0, 0, 0);
          } else {
            const f16x8_t ah = op8(fa[ti][0], fa[ti][1]), al = op8(fa[ti][2], fa[ti][3]);
            const f16x8_t bh = op8(fb[tj][0], fb[tj][1]), bl = op8(fb[tj][2], fb[tj][3]);
            acc1[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc1[ti][tj], 0, 0, 0);
            acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[ti][tj], 0, 0, 0);
            acc1[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc1[ti][tj], 0, 0, 0);
          }
        }
      }
    }
    if (do_bias) {   // column tid % COT, pixel group tid / COT: value = hi + lo / 2048 (scaled)
      const char* stp = ldsb + (i & 1) * STG;
      constexpr int NG = 256 / COT;                 // pixel groups (2 or 4)
      const int c = tid % COT, g = tid / COT;
#pragma unroll
      for (int j = 0; j < P / NG; ++j) {
        const int el = (c >> 5) * (P * 32) + (g * (P / NG) + j) * 32 + (c & 31);
        if constexpr (BF) bsum += __builtin_bit_cast(float, (unsigned)reinterpret_cast<const unsigned short*>(stp)[el] << 16);
        else bsum += (float)reinterpret_cast<const _Float16*>(stp)[el] + (float)reinterpret_cast<const _Float16*>(stp + PLB)[el] * H2_LO_INV;
      }
    }
    if (i + 1 < nst) stash((i + 1) & 1);    // ring slot (i + 1) & 1 was last read in stage i - 1, behind the previous barrier
    __syncthreads();
  }
  if (do_bias) {
    float* red = reinterpret_cast<float*>(ldsb);
    red[tid] = bsum;
    __syncthreads();
    if (tid < COT) {
      float t = 0.f;
#pragma unroll
      for (int g = 0; g < 256 / COT; ++g) t += red[g * COT + tid];
      p.bias_ws[(int64_t)split * p.N + co0 + tid] = t * (1.f / sdy);
    }
  }
  {
    float* o = p.ws + (int64_t)split * p.N * p.K;
    const float ix = 1.f / sx, iy = 1.f / sdy;      // exact powers of two
#pragma unroll
    for (int ti = 0; ti < TCO; ++ti)
#pragma unroll
      for (int tj = 0; tj < TCI; ++tj) {
        const int ci = ci0 + (wci * TCI + tj) * 32 + li;
        if (ci < p.K) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int co = co0 + (wco * TCO + ti) * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
            o[(int64_t)co * p.K + ci] = BF ? acc[ti][tj][e] : (acc[ti][tj][e] + acc1[ti][tj][e] * H2_LO_INV) * ix * iy;
          }
        }
      }
  }
#endif
}

// ---- 3x3 STRIDE-2 weight gradient on the fp16 matrix pipe, fp32 tensors split in flight (round 5; df_conv2d_wgrad_s2_h2) ----------
// The two downsampling layers' weight gradients ran on the fp32 MFMA (wgrad3_ring_kernel<16,2,2>: 113-116 TFLOP/s, 0.68 ms each).
// This is wgrad3_x3_kernel<2>'s scheme at stride 2: a stage = 16 output pixels of one output row = 3 input rows x 33 input columns,
// every staged element split once into two scaled fp16 planes; the input columns are stored DE-INTERLEAVED (17 even slots, then 16
// odd ones, per row and 32-channel half), so that the 16 pixels a tap multiplies -- input columns 2 j + kx - 1 -- are CONSECUTIVE
// 64-byte rows of the transposing-read image exactly as at stride 1 (kx = 0: even slot j, kx = 1: odd slot j, kx = 2: even slot j + 1).
// 12 waves = (32 co x 32 ci quadrant) x kernel row, three taps each; two LDS stages of 2 x 15 KB; the next stage's elements are
// fetched into registers while the current one is multiplied.
// BF (bf16 MFMA mode): one bf16 plane per operand, one MFMA per product, no scales.
template <bool BF>
__global__ __launch_bounds__(768) void wgrad3s2_h2_kernel(WgradParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int P = 16, XW = 34, NE = 17, LC = 64;
  constexpr int YB = 2 * P * 64;                        // dY image bytes of one plane: [2 halves][16 px][64 B]
  constexpr int XH = 3 * XW * 64;                       // one X half: [3 rows][17 even + 16 odd + 1 pad][64 B]
  constexpr int PLB = YB + 2 * XH;                      // one plane of a stage (15104 B)
  constexpr int STG = (BF ? 1 : 2) * PLB;
  constexpr int NYS = YB / 16, NXS = 2 * XH / 16;       // 16-byte slots: 128 + 816
  constexpr int NIT = (NYS + NXS + 767) / 768;          // items per thread (2)
  float sx = 1.f, sdy = 1.f;
  if constexpr (!BF) { sx = df_h2_scale(*p.amax_x); sdy = df_h2_scale(*p.amax_dy); }
  extern __shared__ __attribute__((aligned(16))) char ldsb[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kh = lane >> 5;
  const int quad = wave & 3, ky = wave >> 2;
  const int wci = quad & 1, wco = quad >> 1;
  int bx = blockIdx.x, by = blockIdx.y, split = blockIdx.z;
  if (p.xcd_map) {   // all (ci, co) tiles of a split on one XCD: they read the same x / dy tiles (see wgrad3_ring_kernel)
    const int nt = gridDim.x * gridDim.y;
    const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int lg = df_xcd_swizzle(lin, nt * gridDim.z);
    const int tile = lg % nt;
    split = lg / nt;
    bx = tile % gridDim.x;
    by = tile / gridDim.x;
  }
  const int ci0 = bx * LC, co0 = by * LC;
  const bool do_bias = p.bias_ws && bx == 0;
  float bsum = 0.f;

  f32x16 acc[3], acc1[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc[t][e] = 0.f; acc1[t][e] = 0.f; }

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(p.x.ptr, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.dy.ptr, 0, p.dy_bytes, 0x00020000);
  const int wy = p.dy.w, hx = p.x.h, wx = p.x.w;
  const int c_begin = split * p.chunks_per_split;
  const int c_end = min(c_begin + p.chunks_per_split, p.total_chunks);
  const int nst = max(c_end - c_begin, 0);

  // staging items: item e of a thread = 16-byte slot (tid + 768 e) of a plane image: slots 0 .. 127 = dY ([half][px][4]), the rest
  // = X ([half][row][slot][4]); the source is 8 consecutive fp32 channels (two 16-byte loads)
  unsigned loff[NIT];     // lane-constant source byte offset relative to the stage's row / segment base
  int lpx[NIT], lqy[NIT]; // dY: pixel (lqy = -100); X: input column offset xi - 1, input row offset qy - 1 (off: never in range)
  int ldst[NIT];          // destination byte inside a plane
#pragma unroll
  for (int e = 0; e < NIT; ++e) {
    const int t = tid + 768 * e;
    if (t < NYS) {
      const int h = t >> 6, px = (t & 63) >> 2, q = t & 3;
      lpx[e] = px;
      lqy[e] = -100;
      loff[e] = (unsigned)((px * p.dy.ld + co0 + 32 * h + 8 * q) * 4);
      ldst[e] = 16 * t;
    } else {
      const int s_ = t - NYS;
      const int h = s_ / (3 * XW * 4), r = (s_ - h * 3 * XW * 4) >> 2, q = s_ & 3;
      const int qy = r / XW, pos = r - qy * XW;
      const int xi = pos < NE ? 2 * pos : 2 * (pos - NE) + 1;          // even slots first, then the odd ones (pos 33: padding)
      const bool on = s_ < NXS && pos < 33 && (ci0 + 32 * h + 8 * q) < p.K;
      lpx[e] = xi - 1;
      lqy[e] = on ? qy - 1 : (1 << 28);
      loff[e] = (unsigned)((((qy - 1) * wx + xi - 1) * p.x.ld + ci0 + 32 * h + 8 * q) * 4);
      ldst[e] = s_ < NXS ? YB + 16 * s_ : -1;
    }
  }
  int cur_n, cur_oy, cur_seg;
  {
    const WgChunk c = wg_chunk(p, c_begin < p.total_chunks ? c_begin : 0, P);
    cur_n = c.n; cur_oy = c.oy; cur_seg = c.ox0 / P;
  }
  unsigned yrow = (unsigned)((df_img_base(p.dy, cur_n) + (int64_t)cur_oy * wy * p.dy.ld) * 4);
  unsigned xrow = (unsigned)((df_img_base(p.x, cur_n) + (int64_t)(2 * cur_oy) * wx * p.x.ld) * 4);
  const unsigned yrow_step = (unsigned)(wy * p.dy.ld * 4), xrow_step = (unsigned)(2 * wx * p.x.ld * 4);
  const unsigned yseg_step = (unsigned)(P * p.dy.ld * 4), xseg_step = (unsigned)(2 * P * p.x.ld * 4);
  f32x4 ra[NIT][2];
  auto fetch = [&]() {          // the cursor's chunk -> registers, then advance the cursor
    const int ox0 = cur_seg * P;
    const unsigned ybase = yrow + (unsigned)cur_seg * yseg_step, xbase = xrow + (unsigned)cur_seg * xseg_step;
#pragma unroll
    for (int e = 0; e < NIT; ++e) {
      unsigned v;
      if (lqy[e] == -100) {
        v = (ox0 + lpx[e] < wy) ? ybase + loff[e] : DMA_BAD;
        ra[e][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(yr, v, 0, 0));
        ra[e][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(yr, v + 16, 0, 0));
      } else {
        const bool ok = (unsigned)(2 * cur_oy + lqy[e]) < (unsigned)hx && (unsigned)(2 * ox0 + lpx[e]) < (unsigned)wx;
        v = ok ? xbase + loff[e] : DMA_BAD;
        ra[e][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, v, 0, 0));
        ra[e][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, v + 16, 0, 0));
      }
    }
    if (++cur_seg == p.chunks_per_row) {
      cur_seg = 0;
      yrow += yrow_step;
      xrow += xrow_step;
      if (++cur_oy == p.dy.h) {
        cur_oy = 0;
        ++cur_n;
        yrow = (unsigned)(df_img_base(p.dy, cur_n) * 4);
        xrow = (unsigned)(df_img_base(p.x, cur_n) * 4);
      }
    }
  };
  auto stash = [&](int buf) {   // registers -> (hi, lo) planes -> LDS
    char* st = ldsb + buf * STG;
#pragma unroll
    for (int e = 0; e < NIT; ++e) {
      if (ldst[e] >= 0) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] = ra[e][0][k]; v[4 + k] = ra[e][1][k]; }
        char* d = st + ldst[e];
        if constexpr (BF) {
          *reinterpret_cast<bf16x8_t*>(d) = pack_bf16(v);
        } else {
          f16x8_t h, l;
          df_h2_split(v, lqy[e] == -100 ? sdy : sx, h, l);
          *reinterpret_cast<f16x8_t*>(d) = h;
          *reinterpret_cast<f16x8_t*>(d + PLB) = l;
        }
      }
    }
  };

  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)ldsb;
  const int tr_lane = ((lane & 15) >> 2) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;    // (see wgrad3_tr_kernel)
  const unsigned a_base = lds0 + wco * (P * 64) + (8 * kh) * 64 + tr_lane;
  const unsigned b_base = lds0 + YB + wci * XH + (ky * XW + 8 * kh) * 64 + tr_lane;
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  auto op8 = [](u32x2_t lo, u32x2_t hi) -> f16x8_t {
    u32x4_t v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = hi[0]; v[3] = hi[1];
    return __builtin_bit_cast(f16x8_t, v);
  };

  if (nst > 0) {
    fetch();
    stash(0);
  }
  __syncthreads();
  for (int i = 0; i < nst; ++i) {
    if (i + 1 < nst) fetch();
    const unsigned so = (unsigned)((i & 1) * STG);
    const unsigned aa = a_base + so, ba = b_base + so;
    // all 16 transposing reads of the stage first (dY + three taps, (hi, lo) x (pixels +0..3, +4..7)); offsets: +4 pixels = 256 B,
    // lo plane = +15104 B, tap kx = 1: odd slots = +17 x 64 B, kx = 2: +64 B
    u32x2_t fr[16];
    if constexpr (BF) {
      asm volatile(
          "ds_read_b64_tr_b16 %0, %8\n\t"
          "ds_read_b64_tr_b16 %1, %8 offset:256\n\t"
          "ds_read_b64_tr_b16 %2, %9\n\t"
          "ds_read_b64_tr_b16 %3, %9 offset:256\n\t"
          "ds_read_b64_tr_b16 %4, %9 offset:1088\n\t"
          "ds_read_b64_tr_b16 %5, %9 offset:1344\n\t"
          "ds_read_b64_tr_b16 %6, %9 offset:64\n\t"
          "ds_read_b64_tr_b16 %7, %9 offset:320\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "=&v"(fr[0]), "=&v"(fr[1]), "=&v"(fr[4]), "=&v"(fr[5]), "=&v"(fr[8]), "=&v"(fr[9]), "=&v"(fr[12]), "=&v"(fr[13])
          : "v"(aa), "v"(ba)
          : "memory");
      const bf16x8_t a8 = __builtin_bit_cast(bf16x8_t, op8(fr[0], fr[1]));
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
        acc[kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, __builtin_bit_cast(bf16x8_t, op8(fr[4 + 4 * kx], fr[5 + 4 * kx])), acc[kx], 0, 0, 0);
    } else {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %16\n\t"
        "ds_read_b64_tr_b16 %1, %16 offset:256\n\t"
        "ds_read_b64_tr_b16 %2, %16 offset:15104\n\t"
        "ds_read_b64_tr_b16 %3, %16 offset:15360\n\t"
        "ds_read_b64_tr_b16 %4, %17\n\t"
        "ds_read_b64_tr_b16 %5, %17 offset:256\n\t"
        "ds_read_b64_tr_b16 %6, %17 offset:15104\n\t"
        "ds_read_b64_tr_b16 %7, %17 offset:15360\n\t"
        "ds_read_b64_tr_b16 %8, %17 offset:1088\n\t"
        "ds_read_b64_tr_b16 %9, %17 offset:1344\n\t"
        "ds_read_b64_tr_b16 %10, %17 offset:16192\n\t"
        "ds_read_b64_tr_b16 %11, %17 offset:16448\n\t"
        "ds_read_b64_tr_b16 %12, %17 offset:64\n\t"
        "ds_read_b64_tr_b16 %13, %17 offset:320\n\t"
        "ds_read_b64_tr_b16 %14, %17 offset:15168\n\t"
        "ds_read_b64_tr_b16 %15, %17 offset:15424\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(fr[0]), "=&v"(fr[1]), "=&v"(fr[2]), "=&v"(fr[3]), "=&v"(fr[4]), "=&v"(fr[5]), "=&v"(fr[6]), "=&v"(fr[7]), "=&v"(fr[8]),
          "=&v"(fr[9]), "=&v"(fr[10]), "=&v"(fr[11]), "=&v"(fr[12]), "=&v"(fr[13]), "=&v"(fr[14]), "=&v"(fr[15])
        : "v"(aa), "v"(ba)
        : "memory");
      const f16x8_t ah = op8(fr[0], fr[1]), al = op8(fr[2], fr[3]);
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const f16x8_t bh = op8(fr[4 + 4 * kx], fr[5 + 4 * kx]), bl = op8(fr[6 + 4 * kx], fr[7 + 4 * kx]);
        acc1[kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc1[kx], 0, 0, 0);
        acc[kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[kx], 0, 0, 0);
        acc1[kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc1[kx], 0, 0, 0);
      }
    }
    if (do_bias && tid < 512) {   // column tid & 63, pixel group tid >> 6 (8 groups of 2 pixels): value = hi + lo / 2048 (scaled)
      const char* stp = ldsb + (i & 1) * STG;
      const int c = tid & 63;
#pragma unroll
      for (int j = 0; j < P / 8; ++j) {
        const int el = (c >> 5) * (P * 32) + ((tid >> 6) * (P / 8) + j) * 32 + (c & 31);
        if constexpr (BF) bsum += __builtin_bit_cast(float, (unsigned)reinterpret_cast<const unsigned short*>(stp)[el] << 16);
        else bsum += (float)reinterpret_cast<const _Float16*>(stp)[el] + (float)reinterpret_cast<const _Float16*>(stp + PLB)[el] * H2_LO_INV;
      }
    }
    if (i + 1 < nst) stash((i + 1) & 1);    // ring slot (i + 1) & 1 was last read in stage i - 1, behind the previous barrier
    __syncthreads();
  }
  if (do_bias) {
    float* red = reinterpret_cast<float*>(ldsb);
    if (tid < 512) red[tid] = bsum;
    __syncthreads();
    if (tid < LC) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) t += red[64 * w + tid];
      p.bias_ws[(int64_t)split * p.N + co0 + tid] = t * (1.f / sdy);
    }
  }
  if ((ci0 + wci * 32) < p.K) {
    float* o = p.ws + (int64_t)split * p.N * 9 * p.K;
    const float ix = 1.f / sx, iy = 1.f / sdy;      // exact powers of two
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = co0 + wco * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
        const int ci = ci0 + wci * 32 + li;
        o[((int64_t)co * 9 + ky * 3 + kx) * p.K + ci] = BF ? acc[kx][e] : (acc[kx][e] + acc1[kx][e] * H2_LO_INV) * ix * iy;
      }
  }
#endif
}

// ---- 3x3 stride-1 weight gradient of PRE-SPLIT fp16x2 tensors (round 4; df_conv2d_wgrad_h2p) ------------------------------
// wgrad3_x3_kernel<2> splits every staged fp32 element in registers (global -> registers -> ~7 VALU per element -> LDS, 5.9
// VALU + 1.9 LDS instructions per MFMA, the matrix pipe 47 % busy).  Here BOTH operands arrive split: x and dy are "h2" images
// (df_img.elt = 2) -- per pixel and 32-channel chunk one 128-byte line [32 x fp16 hi | 32 x fp16 lo], the same bytes as the
// fp32 tensor, written by the producing kernel (BatchNorm + GELU passes, conv epilogues, upsample) with the power-of-two scale
// of a bound known BEFORE it writes -- so the stage is pure LDS-DMA (no staging registers, no VALU) through a four-deep ring
// as in wgrad3_tr_kernel, and the loop is transposing reads + MFMAs only.  A 16-byte DMA slot = 8 channels of one plane of one
// pixel; the LDS image per plane is wgrad3_tr_kernel's ([half][pixel][64 B]: dY 4 KB, X 2 x 3 x 34 x 64 B padded to 13 KB),
// two planes per stage = 34 one-KB ops over 12 waves = 3 per wave (the two spare slots repeat ops 0 / 1: same bytes to the same
// place, so that every wave issues the same count and the waits are compile-time constants).
// PIPE (round 6; D == 4 only): the transposing reads run HALF A STAGE AHEAD of the products -- the fragments of (stage i, step 0) are
// read during the products of (stage i - 1, step 1), i.e. BEFORE barrier i, and those of (i, 1) under the products of (i, 0) -- so a
// wave leaves every barrier with a product ready to issue.  In the plain form all 12 waves leave the barrier into 32 reads each
// (768 LDS cycles per stage for the CU) and the matrix pipes drain once per 32-pixel stage.  The price: the wave's share of stage
// i + 1 (not i) must have landed at barrier i, so two stages (not three) of DMA are in flight behind the one being read.  Same
// products in the same order: bit-identical to the plain form.
template <int D, bool PIPE = false>
__global__ __launch_bounds__(768) void wgrad3_h2p_kernel(WgradParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int P = 32, XW = P + 2, LC = 64;
  constexpr int YB = 2 * P * 64;                        // dY image bytes of one plane: [2 halves][32 px][64 B]   = 4096
  constexpr int XH = 3 * XW * 64;                       // one X half: [3 rows x 34 px][64 B]                     = 6528
  constexpr int NYO = YB / 1024, NXO = (2 * XH + 1023) / 1024;   // 1-KB DMA ops per plane: 4 + 13
  constexpr int PLB = (NYO + NXO) * 1024;               // plane stride inside a stage (17408)
  constexpr int STG = 2 * PLB;                          // stage bytes (34816)
  constexpr int NOPS = 2 * (NYO + NXO);                 // 34 real ops per stage; 12 waves x 3
  static_assert(D >= 2 && D <= 4 && NOPS <= 36 && PLB == 17408, "ring");
  const float sx = df_h2_scale(*p.amax_x), sdy = df_h2_scale(*p.amax_dy);
  extern __shared__ __attribute__((aligned(16))) char ldsb[];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kh = lane >> 5;
  const int quad = wave & 3, ky = wave >> 2;
  const int wci = quad & 1, wco = quad >> 1;
  int bx = blockIdx.x, by = blockIdx.y, split = blockIdx.z;
  if (p.xcd_map) {   // all (ci, co) tiles of a split on one XCD: they read the same x / dy tiles (see wgrad3_ring_kernel)
    const int nt = gridDim.x * gridDim.y;
    const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int lg = df_xcd_swizzle(lin, nt * gridDim.z);
    const int tile = lg % nt;
    split = lg / nt;
    bx = tile % gridDim.x;
    by = tile / gridDim.x;
  }
  const int ci0 = bx * LC, co0 = by * LC;
  const bool do_bias = p.bias_ws && bx == 0;
  float bsum = 0.f;

  f32x16 acc[3], acc1[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc[t][e] = 0.f; acc1[t][e] = 0.f; }

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(p.x.ptr, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.dy.ptr, 0, p.dy_bytes, 0x00020000);
  const int wy = p.dy.w, hx = p.x.h, wx = p.x.w;
  const int c_begin = split * p.chunks_per_split;
  const int c_end = min(c_begin + p.chunks_per_split, p.total_chunks);
  const int nst = max(c_end - c_begin, 0);

  // DMA ops of this wave: op j = (wave + 12 i) mod 34 -> plane j / 17, op-in-plane r = j % 17 (r < 4: dY, else X); lane-constant
  // source bytes relative to the stage's row / segment base, wave-uniform cursor for the rest (no divisions in the loop)
  // (lpq packs the lane's pixel and row offsets of an op -- (qy + 1) << 8 | (px + 1), "off" rows as a huge qy -- into ONE register:
  //  at 168 registers (three 12-wave... waves per SIMD) the separate lpx / lqy arrays spilled, and the reload inside issue() came with an
  //  s_waitcnt vmcnt(0) that drained the whole DMA ring once per stage on 8 of the 12 waves)
  bool op_y[3];
  unsigned loff[3];
  int lpq[3], ldst[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int j = (wave + 12 * i) % NOPS;
    const int pl = j / (NYO + NXO), r = j - pl * (NYO + NXO);
    op_y[i] = r < NYO;
    ldst[i] = pl * PLB + r * 1024;
    if (r < NYO) {
      const int s = r * 64 + lane;                       // 16-byte slot of the dY image: [half][px][4]
      const int h = s >> 7, px = (s & 127) >> 2, q = s & 3;
      lpq[i] = (1 << 8) | (px + 1);
      loff[i] = (unsigned)((px * p.dy.ld + co0 + 32 * h) * 4 + pl * 64 + q * 16);
    } else {
      const int s = (r - NYO) * 64 + lane;               // slot of the X image: [half][row qy][xi][4 slots]
      const int h = s / (3 * XW * 4), rr = (s - h * 3 * XW * 4) >> 2, q = s & 3;
      const int qy = rr / XW, xi = rr - qy * XW;
      const bool on = h < 2 && (ci0 + 32 * h) < p.K;
      lpq[i] = ((on ? qy : (1 << 20)) << 8) | xi;
      loff[i] = (unsigned)((((qy - 1) * wx + xi - 1) * p.x.ld + ci0 + 32 * h) * 4 + pl * 64 + q * 16);
    }
  }
  // PIPE: the six lane constants live in LDS behind the ring ([6][768] words, a lane reads back what it wrote): with both fragment
  // sets alive across issue() they do not fit the 168 registers, and a spilled constant comes back through scratch with an
  // s_waitcnt vmcnt(0) -- i.e. a drain of the DMA ring inside the loop
  unsigned* const optab = reinterpret_cast<unsigned*>(ldsb + D * STG);
  if constexpr (PIPE) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      optab[(2 * i) * 768 + tid] = loff[i];
      optab[(2 * i + 1) * 768 + tid] = (unsigned)lpq[i];
    }
  }
  // COLUMN-MAJOR chunk order (round 4): chunk index -> (image n, segment seg, output row oy) with oy running fastest, so that
  // consecutive stages of a workgroup read the x rows (oy - 1, oy, oy + 1), (oy, oy + 1, oy + 2), ...: two of the three rows of a stage
  // were fetched one stage earlier and are still in L2.  Walking along the row first (the other weight-gradient kernels' order) put a
  // whole row walk -- 16 stages x 35 KB x 32 workgroups per XCD -- between the three uses of a row: x came from HBM three times
  // (FETCH_SIZE: 1.41 GB per launch against 0.88 GB of operands).  Only the order of the fp32 accumulation over pixels changes.
  int cur_n, cur_oy, cur_seg;
  {
    const int ch = c_begin < p.total_chunks ? c_begin : 0;
    const int per_img = p.chunks_per_row * p.dy.h;
    cur_n = ch / per_img;
    const int rem = ch - cur_n * per_img;
    cur_seg = rem / p.dy.h;
    cur_oy = rem - cur_seg * p.dy.h;
  }
  unsigned yrow = (unsigned)((df_img_base(p.dy, cur_n) + (int64_t)cur_oy * wy * p.dy.ld) * 4);
  unsigned xrow = (unsigned)((df_img_base(p.x, cur_n) + (int64_t)cur_oy * wx * p.x.ld) * 4);
  const unsigned yrow_step = (unsigned)(wy * p.dy.ld * 4), xrow_step = (unsigned)(wx * p.x.ld * 4);
  const unsigned yseg_step = (unsigned)(P * p.dy.ld * 4), xseg_step = (unsigned)(P * p.x.ld * 4);
  auto issue = [&](int buf) {   // loads the cursor's chunk into ring slot `buf`, then advances the cursor
    char* slot = ldsb + buf * STG;
    const int ox0 = cur_seg * P;
    const unsigned ybase = yrow + (unsigned)cur_seg * yseg_step, xbase = xrow + (unsigned)cur_seg * xseg_step;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int lpq_i = PIPE ? (int)optab[(2 * i + 1) * 768 + tid] : lpq[i];
      const unsigned loff_i = PIPE ? optab[(2 * i) * 768 + tid] : loff[i];
      const int lpx = (lpq_i & 255) - 1, lqy = (lpq_i >> 8) - 1;
      if constexpr (PIPE) {
        // branch-free (round 6): the op's tensor (dY or x) picked by scalar selects -- as wave-uniform branches the three ops cost
        // ~12 taken branches per stage and wave; a dY op has row offset 0 and an unbounded row range
        const bool y = op_y[i];
        const unsigned hb = y ? 0x7fffffffu : (unsigned)hx, wb = y ? (unsigned)wy : (unsigned)wx;
        const unsigned base = y ? ybase : xbase;
        const bool ok = (unsigned)(cur_oy + lqy) < hb && (unsigned)(ox0 + lpx) < wb;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(y ? yr : xr, (lds_ptr_t)(slot + ldst[i]), 16, ok ? base + loff_i : DMA_BAD, 0, 0, 0);
      } else if (op_y[i]) {                              // (wave-uniform)
        const bool ok = ox0 + lpx < wy;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(yr, (lds_ptr_t)(slot + ldst[i]), 16, ok ? ybase + loff_i : DMA_BAD, 0, 0, 0);
      } else {
        const bool ok = (unsigned)(cur_oy + lqy) < (unsigned)hx && (unsigned)(ox0 + lpx) < (unsigned)wx;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr_t)(slot + ldst[i]), 16, ok ? xbase + loff_i : DMA_BAD, 0, 0, 0);
      }
    }
    yrow += yrow_step;
    xrow += xrow_step;
    if (++cur_oy == p.dy.h) {     // next column segment of the image, or the next image (its base need not follow the previous one)
      cur_oy = 0;
      if (++cur_seg == p.chunks_per_row) {
        cur_seg = 0;
        ++cur_n;
      }
      yrow = (unsigned)(df_img_base(p.dy, cur_n) * 4);
      xrow = (unsigned)(df_img_base(p.x, cur_n) * 4);
    }
  };

  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)ldsb;
  const int tr_lane = ((lane & 15) >> 2) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;    // (see wgrad3_tr_kernel)
  const unsigned a_base = lds0 + wco * (P * 64) + (8 * kh) * 64 + tr_lane;
  const unsigned b_base = lds0 + YB + wci * XH + (ky * XW + 8 * kh) * 64 + tr_lane;
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  auto op8 = [](u32x2_t lo, u32x2_t hi) -> f16x8_t {
    u32x4_t v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = hi[0]; v[3] = hi[1];
    return __builtin_bit_cast(f16x8_t, v);
  };
#pragma unroll
  for (int d = 0; d < D - 1; ++d)
    if (d < nst) issue(d);
  if constexpr (PIPE) {
    static_assert(D == 4, "PIPE: four-deep ring");
    u32x2_t f0[16], f1[16];
    // one 16-pixel step's operands: dY (hi: %0 %1, lo: %2 %3), then the three taps of x (hi, hi, lo, lo each)
#define DF_H2P_READ16(F, AA, BA)                                                                                            \
    asm volatile(                                                                                                            \
        "ds_read_b64_tr_b16 %0, %16\n\t"                                                                                     \
        "ds_read_b64_tr_b16 %1, %16 offset:256\n\t"                                                                          \
        "ds_read_b64_tr_b16 %2, %16 offset:17408\n\t"                                                                        \
        "ds_read_b64_tr_b16 %3, %16 offset:17664\n\t"                                                                        \
        "ds_read_b64_tr_b16 %4, %17\n\t"                                                                                     \
        "ds_read_b64_tr_b16 %5, %17 offset:256\n\t"                                                                          \
        "ds_read_b64_tr_b16 %6, %17 offset:17408\n\t"                                                                        \
        "ds_read_b64_tr_b16 %7, %17 offset:17664\n\t"                                                                        \
        "ds_read_b64_tr_b16 %8, %17 offset:64\n\t"                                                                           \
        "ds_read_b64_tr_b16 %9, %17 offset:320\n\t"                                                                          \
        "ds_read_b64_tr_b16 %10, %17 offset:17472\n\t"                                                                       \
        "ds_read_b64_tr_b16 %11, %17 offset:17728\n\t"                                                                       \
        "ds_read_b64_tr_b16 %12, %17 offset:128\n\t"                                                                         \
        "ds_read_b64_tr_b16 %13, %17 offset:384\n\t"                                                                         \
        "ds_read_b64_tr_b16 %14, %17 offset:17536\n\t"                                                                       \
        "ds_read_b64_tr_b16 %15, %17 offset:17792"                                                                           \
        : "=&v"(F[0]), "=&v"(F[1]), "=&v"(F[2]), "=&v"(F[3]), "=&v"(F[4]), "=&v"(F[5]), "=&v"(F[6]), "=&v"(F[7]), "=&v"(F[8]),  \
          "=&v"(F[9]), "=&v"(F[10]), "=&v"(F[11]), "=&v"(F[12]), "=&v"(F[13]), "=&v"(F[14]), "=&v"(F[15])                      \
        : "v"(AA), "v"(BA)                                                                                                    \
        : "memory")
    // every read issued so far has landed; the registers ride as read-write operands so that no product (and no copy the register
    // allocator may want at the loop's back edge) can be placed in front of the wait that covers them
#define DF_H2P_WAIT16(F)                                                                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                                      \
                 : "+v"(F[0]), "+v"(F[1]), "+v"(F[2]), "+v"(F[3]), "+v"(F[4]), "+v"(F[5]), "+v"(F[6]), "+v"(F[7]), "+v"(F[8]),  \
                   "+v"(F[9]), "+v"(F[10]), "+v"(F[11]), "+v"(F[12]), "+v"(F[13]), "+v"(F[14]), "+v"(F[15])                    \
                 :: "memory")
    auto products = [&](const u32x2_t (&F)[16]) {
      const f16x8_t ah = op8(F[0], F[1]), al = op8(F[2], F[3]);
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const f16x8_t bh = op8(F[4 + 4 * kx], F[5 + 4 * kx]), bl = op8(F[6 + 4 * kx], F[7 + 4 * kx]);
        acc1[kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc1[kx], 0, 0, 0);
        acc[kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[kx], 0, 0, 0);
        acc1[kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc1[kx], 0, 0, 0);
      }
    };
    if (nst > 0) {
      switch (min(D - 2, nst - 1)) {          // this wave's share of stage 0 has landed ...
        case 2: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // ... and everyone's
      DF_H2P_READ16(f0, a_base, b_base);
      DF_H2P_WAIT16(f0);
    }
    // (the x operand's address as the dY operand's + a wave-uniform distance, formed where it is used: one lane register less
    //  across the loop -- at 168 registers that is the difference between no scratch and a reload + vmcnt(0) per stage)
    const unsigned b_delta0 = __builtin_amdgcn_readfirstlane((unsigned)(YB + wci * XH + ky * XW * 64 - wco * (P * 64)));
    for (int i = 0; i < nst; ++i) {
      // this wave's share of stage i + 1 has landed (stage i + 2, if there is one, may be in flight) ...
      if (i + 2 < nst) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // ... and everyone's; every wave has also finished reading ring slot (i - 1) % D
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      const unsigned so = (unsigned)((i % D) * STG) + 16 * 64;
      unsigned b_delta = b_delta0;
      asm volatile("" : "+s"(b_delta));               // (not hoistable)
      const unsigned aa1 = a_base + so, ba1 = aa1 + b_delta;
      DF_H2P_READ16(f1, aa1, ba1);                    // (stage i, step 1) under the products of (i, 0)
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);   // (+1.5 %: a wave that has its fragments keeps the pipe against its two neighbours' scalar / DMA code)
      products(f0);
      __builtin_amdgcn_s_setprio(0);
      DF_H2P_WAIT16(f1);
      // the DMA of stage i + 3 into the slot stage i - 1 left behind barrier i -- issued HERE, where f0 is dead: its address
      // arithmetic beside both fragment sets does not fit the 168 registers of three waves per SIMD
      if (i + D - 1 < nst) issue((i + D - 1) % D);
      __builtin_amdgcn_sched_barrier(0);
      if (i + 1 < nst) {
        const unsigned sn = (unsigned)(((i + 1) % D) * STG);
        const unsigned aa0 = a_base + sn, ba0 = aa0 + b_delta;
        DF_H2P_READ16(f0, aa0, ba0);                  // (stage i + 1, step 0) under the products of (i, 1)
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);   // (+1.5 %: a wave that has its fragments keeps the pipe against its two neighbours' scalar / DMA code)
      products(f1);
      __builtin_amdgcn_s_setprio(0);
      if (do_bias && tid < 512) {   // column tid & 63, pixel group tid >> 6 (8 groups of 4 pixels): value = hi + lo / 2048 (scaled)
        const char* stp = ldsb + (i % D) * STG;
        // the lane's offset re-derived per stage from the hardware lane id (a handful of VALU): kept across the loop it is one
        // register too many, and a reload from scratch comes with an s_waitcnt vmcnt(0) -- a drain of the DMA ring per stage
        const int t2 = wave * 64 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        const int c = t2 & 63;
#pragma unroll
        for (int j = 0; j < P / 8; ++j) {
          const int el = (c >> 5) * (P * 32) + ((t2 >> 6) * (P / 8) + j) * 32 + (c & 31);
          bsum += (float)reinterpret_cast<const _Float16*>(stp)[el] + (float)reinterpret_cast<const _Float16*>(stp + PLB)[el] * H2_LO_INV;
        }
      }
      if (i + 1 < nst) DF_H2P_WAIT16(f0);             // (landed long ago: the products above took ~600 cycles)
    }
#undef DF_H2P_READ16
#undef DF_H2P_WAIT16
  } else
  for (int i = 0; i < nst; ++i) {
    // this wave's DMA share of stage i has landed (the ops of up to D - 2 later stages may still be in flight) ...
    switch (min(D - 2, nst - 1 - i)) {
      case 2: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
    // ... and everyone's; every wave has also finished reading ring slot (i - 1) % D (raw barrier: see wgrad3_tr_kernel)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (i + D - 1 < nst) issue((i + D - 1) % D);
    const unsigned so = (unsigned)((i % D) * STG);
    // The reads of BOTH 16-pixel steps are issued before the first product (two steps x (dY + 3 taps of x) x (hi, lo) x 2 halves of
    // the 8-deep operand = 32 transposing reads; lgkmcnt is a 4-bit counter, so the first step's 16 + 12 of the second go out, the
    // wait lgkmcnt(12) covers the first step -- LDS returns in order -- and the last 4 follow): the second step's reads land under
    // the first step's products -- one exposed LDS latency per stage instead of eight.  The waits carry the registers as
    // read-write operands so that the compiler cannot move a product in front of the wait that covers its operands.
    u32x2_t fr[2][16];
    const unsigned aa0 = a_base + so, ba0 = b_base + so, aa1 = aa0 + 16 * 64, ba1 = ba0 + 16 * 64;
    asm volatile(
        "ds_read_b64_tr_b16 %0, %16\n\t"
        "ds_read_b64_tr_b16 %1, %16 offset:256\n\t"
        "ds_read_b64_tr_b16 %2, %16 offset:17408\n\t"
        "ds_read_b64_tr_b16 %3, %16 offset:17664\n\t"
        "ds_read_b64_tr_b16 %4, %17\n\t"
        "ds_read_b64_tr_b16 %5, %17 offset:256\n\t"
        "ds_read_b64_tr_b16 %6, %17 offset:17408\n\t"
        "ds_read_b64_tr_b16 %7, %17 offset:17664\n\t"
        "ds_read_b64_tr_b16 %8, %17 offset:64\n\t"
        "ds_read_b64_tr_b16 %9, %17 offset:320\n\t"
        "ds_read_b64_tr_b16 %10, %17 offset:17472\n\t"
        "ds_read_b64_tr_b16 %11, %17 offset:17728\n\t"
        "ds_read_b64_tr_b16 %12, %17 offset:128\n\t"
        "ds_read_b64_tr_b16 %13, %17 offset:384\n\t"
        "ds_read_b64_tr_b16 %14, %17 offset:17536\n\t"
        "ds_read_b64_tr_b16 %15, %17 offset:17792"
        : "=&v"(fr[0][0]), "=&v"(fr[0][1]), "=&v"(fr[0][2]), "=&v"(fr[0][3]), "=&v"(fr[0][4]), "=&v"(fr[0][5]), "=&v"(fr[0][6]),
          "=&v"(fr[0][7]), "=&v"(fr[0][8]), "=&v"(fr[0][9]), "=&v"(fr[0][10]), "=&v"(fr[0][11]), "=&v"(fr[0][12]),
          "=&v"(fr[0][13]), "=&v"(fr[0][14]), "=&v"(fr[0][15])
        : "v"(aa0), "v"(ba0)
        : "memory");
    asm volatile(
        "ds_read_b64_tr_b16 %0, %12\n\t"
        "ds_read_b64_tr_b16 %1, %12 offset:256\n\t"
        "ds_read_b64_tr_b16 %2, %12 offset:17408\n\t"
        "ds_read_b64_tr_b16 %3, %12 offset:17664\n\t"
        "ds_read_b64_tr_b16 %4, %13\n\t"
        "ds_read_b64_tr_b16 %5, %13 offset:256\n\t"
        "ds_read_b64_tr_b16 %6, %13 offset:17408\n\t"
        "ds_read_b64_tr_b16 %7, %13 offset:17664\n\t"
        "ds_read_b64_tr_b16 %8, %13 offset:64\n\t"
        "ds_read_b64_tr_b16 %9, %13 offset:320\n\t"
        "ds_read_b64_tr_b16 %10, %13 offset:17472\n\t"
        "ds_read_b64_tr_b16 %11, %13 offset:17728"
        : "=&v"(fr[1][0]), "=&v"(fr[1][1]), "=&v"(fr[1][2]), "=&v"(fr[1][3]), "=&v"(fr[1][4]), "=&v"(fr[1][5]), "=&v"(fr[1][6]),
          "=&v"(fr[1][7]), "=&v"(fr[1][8]), "=&v"(fr[1][9]), "=&v"(fr[1][10]), "=&v"(fr[1][11])
        : "v"(aa1), "v"(ba1)
        : "memory");
    asm volatile("s_waitcnt lgkmcnt(12)"
                 : "+v"(fr[0][0]), "+v"(fr[0][1]), "+v"(fr[0][2]), "+v"(fr[0][3]), "+v"(fr[0][4]), "+v"(fr[0][5]), "+v"(fr[0][6]), "+v"(fr[0][7]),
                   "+v"(fr[0][8]), "+v"(fr[0][9]), "+v"(fr[0][10]), "+v"(fr[0][11]), "+v"(fr[0][12]), "+v"(fr[0][13]), "+v"(fr[0][14]),
                   "+v"(fr[0][15])
                 :: "memory");
    asm volatile(
        "ds_read_b64_tr_b16 %0, %4 offset:128\n\t"
        "ds_read_b64_tr_b16 %1, %4 offset:384\n\t"
        "ds_read_b64_tr_b16 %2, %4 offset:17536\n\t"
        "ds_read_b64_tr_b16 %3, %4 offset:17792"
        : "=&v"(fr[1][12]), "=&v"(fr[1][13]), "=&v"(fr[1][14]), "=&v"(fr[1][15])
        : "v"(ba1)
        : "memory");
#pragma unroll
    for (int ks = 0; ks < P / 16; ++ks) {
      if (ks == 1) {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(fr[1][0]), "+v"(fr[1][1]), "+v"(fr[1][2]), "+v"(fr[1][3]), "+v"(fr[1][4]), "+v"(fr[1][5]), "+v"(fr[1][6]), "+v"(fr[1][7]),
                       "+v"(fr[1][8]), "+v"(fr[1][9]), "+v"(fr[1][10]), "+v"(fr[1][11]), "+v"(fr[1][12]), "+v"(fr[1][13]), "+v"(fr[1][14]),
                       "+v"(fr[1][15])
                     :: "memory");
      }
      const f16x8_t ah = op8(fr[ks][0], fr[ks][1]), al = op8(fr[ks][2], fr[ks][3]);
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const f16x8_t bh = op8(fr[ks][4 + 4 * kx], fr[ks][5 + 4 * kx]), bl = op8(fr[ks][6 + 4 * kx], fr[ks][7 + 4 * kx]);
        acc1[kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc1[kx], 0, 0, 0);
        acc[kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[kx], 0, 0, 0);
        acc1[kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc1[kx], 0, 0, 0);
      }
    }
    if (do_bias && tid < 512) {   // column tid & 63, pixel group tid >> 6 (8 groups of 4 pixels): value = hi + lo / 2048 (scaled)
      const char* stp = ldsb + (i % D) * STG;
      const int c = tid & 63;
#pragma unroll
      for (int j = 0; j < P / 8; ++j) {
        const int el = (c >> 5) * (P * 32) + ((tid >> 6) * (P / 8) + j) * 32 + (c & 31);
        bsum += (float)reinterpret_cast<const _Float16*>(stp)[el] + (float)reinterpret_cast<const _Float16*>(stp + PLB)[el] * H2_LO_INV;
      }
    }
  }
  if (do_bias) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(ldsb);
    if (tid < 512) red[tid] = bsum;
    __syncthreads();
    if (tid < LC) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) t += red[64 * w + tid];
      p.bias_ws[(int64_t)split * p.N + co0 + tid] = t * (1.f / sdy);
    }
  }
  if ((ci0 + wci * 32) < p.K) {
    float* o = p.ws + (int64_t)split * p.N * 9 * p.K;
    const float ix = 1.f / sx, iy = 1.f / sdy;      // exact powers of two
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = co0 + wco * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
        const int ci = ci0 + wci * 32 + li;
        o[((int64_t)co * 9 + ky * 3 + kx) * p.K + ci] = (acc[kx][e] + acc1[kx][e] * H2_LO_INV) * ix * iy;
      }
  }
#endif
}

template <int CIT>
__global__ __launch_bounds__(256, 2) void wgrad1x1_dma_kernel(WgradParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int P = 32, COT = 128;
  constexpr int TCI = CIT / 64;
  constexpr int YSZ = P * COT, XSZ = P * CIT;
  constexpr int YPP = 1024 / (COT * 4), XPP = 1024 / (CIT * 4);   // pixels per 1-KB DMA op (2; 2 or 4)
  constexpr int NYI = P / YPP, NXI = P / XPP;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* dYs = lds;
  float* Xs = lds + 2 * YSZ;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kh = lane >> 5;
  const int wco = wave & 1, wci = wave >> 1;
  const int ci0 = blockIdx.x * CIT, co0 = blockIdx.y * COT, split = blockIdx.z;
  const int ciw = ci0 + wci * (CIT / 2);

  f32x16 acc[2][TCI];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TCI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(p.x.ptr, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.dy.ptr, 0, p.dy_bytes, 0x00020000);
  const int wy = p.dy.w;
  const int c_begin = split * p.chunks_per_split;
  const int c_end = min(c_begin + p.chunks_per_split, p.total_chunks);
  const int ylp = lane / (COT / 4), yc4 = lane % (COT / 4);
  const int xlp = lane / (CIT / 4), xc4 = lane % (CIT / 4);
  const bool ci_ok = ci0 + xc4 * 4 < p.K;

  auto issue = [&](int ch, int buf) {
    const WgChunk c = wg_chunk(p, ch, P);
    const int64_t yb = df_img_base(p.dy, c.n) + (int64_t)c.oy * wy * p.dy.ld + co0 + yc4 * 4;
    const int64_t xb = df_img_base(p.x, c.n) + (int64_t)c.oy * wy * p.x.ld + ci0 + xc4 * 4;
    for (int k = wave; k < NYI; k += 4) {
      const int px = c.ox0 + YPP * k + ylp;
      const bool ok = px < wy && wg_row_ok(p, px);
      const unsigned vo = ok ? (unsigned)((yb + (int64_t)px * p.dy.ld) * 4) : DMA_BAD;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(yr, (lds_ptr_t)(dYs + buf * YSZ + k * 256), 16, vo, 0, 0, 0);
    }
    for (int k = wave; k < NXI; k += 4) {
      const int px = c.ox0 + XPP * k + xlp;
      const bool ok = ci_ok && px < wy && wg_row_ok(p, px);
      const unsigned vo = ok ? (unsigned)((xb + (int64_t)px * p.x.ld) * 4) : DMA_BAD;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr_t)(Xs + buf * XSZ + k * 256), 16, vo, 0, 0, 0);
    }
  };
  if (c_begin < c_end) issue(c_begin, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const bool wave_active = ciw < p.K;
  for (int ch = c_begin; ch < c_end; ++ch) {
    const int buf = (ch - c_begin) & 1;
    if (ch + 1 < c_end) issue(ch + 1, buf ^ 1);
    if (wave_active) {
      const float* dyb = dYs + buf * YSZ;
      const float* xbuf = Xs + buf * XSZ;
#pragma unroll 4
      for (int ks = 0; ks < P / 2; ++ks) {
        const int px = 2 * ks + kh;
        float a[2], b[TCI];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = dyb[px * COT + wco * 64 + i * 32 + li];
#pragma unroll
        for (int j = 0; j < TCI; ++j) b[j] = xbuf[px * CIT + wci * (CIT / 2) + j * 32 + li];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < TCI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (wave_active) {
    float* o = p.ws + (int64_t)split * p.N * p.K;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < TCI; ++j) {
        const int ci = ciw + j * 32 + li;
        if (ci < p.K) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int co = co0 + wco * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
            o[(int64_t)co * p.K + ci] = acc[i][j][e];
          }
        }
      }
  }
#endif
}

template <typename K>
static int launch_wgrad_dma(K kern, dim3 grid, size_t lds_bytes, hipStream_t s, const WgradParams& p, int threads = 256) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds_bytes);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, grid, dim3(threads), lds_bytes, s, p);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

__global__ void wgrad_reduce_kernel(const float* __restrict__ ws, int splits, int64_t per_split, int row_len,
                                    float* __restrict__ dw, int64_t ld_co, int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= per_split) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int k = 0;
  for (; k + 4 <= splits; k += 4) {
    s0 += ws[(int64_t)k * per_split + i];
    s1 += ws[(int64_t)(k + 1) * per_split + i];
    s2 += ws[(int64_t)(k + 2) * per_split + i];
    s3 += ws[(int64_t)(k + 3) * per_split + i];
  }
  for (; k < splits; ++k) s0 += ws[(int64_t)k * per_split + i];
  const float s = (s0 + s1) + (s2 + s3);
  const int64_t co = i / row_len, r = i - co * row_len;
  float* o = dw + co * ld_co + r;
  *o = accumulate ? (*o + s) : s;
}

// ... and the bias partials of the same split-K pass in the SAME launch (round 4, second session: a conv layer's weight gradient
// was followed by wgrad_reduce + colsum_finalize, two 5-25 us launches; 33 of each per step).  Blocks [0, nblk_w) reduce the weight
// partials exactly as above; the blocks behind them sum bias_ws [splits][cout] over the splits in double (32 columns x 8 split lanes each).
__global__ void wgrad_reduce_bias_kernel(const float* __restrict__ ws, int splits, int64_t per_split, int row_len,
                                         float* __restrict__ dw, int64_t ld_co, int accumulate, int nblk_w,
                                         const float* __restrict__ bias_ws, int cout, float* __restrict__ db) {
  if ((int)blockIdx.x >= nblk_w) {   // 32 columns x 8 split lanes per block (256 threads), as colsum_finalize_kernel
    __shared__ double red[8][32];
    const int cl = threadIdx.x & 31, tl = threadIdx.x >> 5;
    const int c = ((int)blockIdx.x - nblk_w) * 32 + cl;
    double s = 0.0;
    if (c < cout) {
      int k = tl;
      for (; k + 24 < splits; k += 32) {   // four loads in flight
        const float v0 = bias_ws[(int64_t)k * cout + c], v1 = bias_ws[(int64_t)(k + 8) * cout + c];
        const float v2 = bias_ws[(int64_t)(k + 16) * cout + c], v3 = bias_ws[(int64_t)(k + 24) * cout + c];
        s += (double)v0; s += (double)v1; s += (double)v2; s += (double)v3;
      }
      for (; k < splits; k += 8) s += (double)bias_ws[(int64_t)k * cout + c];
    }
    red[tl][cl] = s;
    __syncthreads();
    if (tl == 0 && c < cout) {
      for (int j = 1; j < 8; ++j) s += red[j][cl];
      db[c] = (float)s;
    }
    return;
  }
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= per_split) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int k = 0;
  for (; k + 4 <= splits; k += 4) {
    s0 += ws[(int64_t)k * per_split + i];
    s1 += ws[(int64_t)(k + 1) * per_split + i];
    s2 += ws[(int64_t)(k + 2) * per_split + i];
    s3 += ws[(int64_t)(k + 3) * per_split + i];
  }
  for (; k < splits; ++k) s0 += ws[(int64_t)k * per_split + i];
  const float s = (s0 + s1) + (s2 + s3);
  const int64_t co = i / row_len, r = i - co * row_len;
  float* o = dw + co * ld_co + r;
  *o = accumulate ? (*o + s) : s;
}

__global__ void weight_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int cout, int taps,
                                        int cin) {
  // wt[ci][t][co] = w[co][t][ci]; one thread per output element (co fastest => coalesced writes)
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)cout * taps * cin;
  if (i >= total) return;
  const int co = (int)(i % cout);
  const int64_t q = i / cout;
  const int t = (int)(q % taps), ci = (int)(q / taps);
  wt[i] = w[((int64_t)co * taps + t) * cin + ci];
}

}  // namespace

static inline bool wgrad_use_1x1(int ksize, int cout) { return ksize == 1 && (cout % 128) == 0; }
static inline int wgrad_cit(int cin) { return cin >= 128 ? 128 : 64; }
static inline int wgrad_ring_depth(int ksize, int stride) {   // 12-wave kernel: 0 = off, 3 = one workgroup per CU with a
  static const int ring = getenv("DF_WGRAD_RING") ? atoi(getenv("DF_WGRAD_RING")) : 2;   // 3-deep ring, 2 = two per CU
  // (stride 2: wgrad_ring_s2() below)
  return (ksize == 3 && stride == 1) ? ring : 0;
}
// stride-2 3x3 weight gradient on the 12-wave ring kernel too (16-pixel chunks: 3 x 33 input pixels per stage, 29 KB, two
// workgroups per CU).  Round 1 measured this form SLOWER than the register-prefetch kernel (3.06 vs 2.41 ms per step); with the
// ring kernel's strength-reduced DMA addressing it is faster: fp32 87 -> 97 and 99 -> 110 TFLOP/s on the two layers, bf16
// mode 78 -> 254 and 80 -> 284 (tools/ab_wgrad_s2.py).  DF_WGRAD_RING_S2=0: the register-prefetch kernel (A/B, tests).
static inline int wgrad_ring_s2() {
  static const int on = getenv("DF_WGRAD_RING_S2") ? atoi(getenv("DF_WGRAD_RING_S2")) : 1;
  return on;
}
static inline int wgrad_chunk(int ksize, int stride) {   // output pixels per chunk
  return (ksize == 3 && stride == 2 && wgrad_ring_s2()) ? 16 : 32;
}

extern "C" int df_conv2d_wgrad_splits(df_img x, df_img dy, int ksize, int stride) {
  const int P = wgrad_chunk(ksize, stride);
  const int tiles = wgrad_use_1x1(ksize, dy.c) ? ((x.c + wgrad_cit(x.c) - 1) / wgrad_cit(x.c)) * (dy.c / 128)
                                               : ((x.c + 63) / 64) * (dy.c / 64);
  const int64_t chunks = (int64_t)dy.n * dy.h * ((dy.w + P - 1) / P);
  static const int target = getenv("DF_WGRAD_BLOCKS") ? atoi(getenv("DF_WGRAD_BLOCKS")) : 512;  // = one resident round at 2 workgroups per CU (1024: +0.6 ms of partial-sum traffic, 768: a ragged second round)
  const int tgt = wgrad_ring_depth(ksize, stride) == 3 ? target / 2 : target;   // one resident workgroup per CU
  int64_t splits = (tgt + tiles - 1) / tiles;
  if (splits > chunks) splits = chunks;
  if (splits < 1) splits = 1;
  // make every split non-empty
  const int64_t cps = (chunks + splits - 1) / splits;
  splits = (chunks + cps - 1) / cps;
  return (int)splits;
}

extern "C" int df_conv2d_wgrad(df_img x, df_img dy, int ksize, int stride, int pad, float* ws, int splits,
                               const int32_t* row_counts, int rows_per_seg, float* bias_ws, void* stream) {
  return df_conv2d_wgrad_mp(x, dy, ksize, stride, pad, ws, splits, row_counts, rows_per_seg, bias_ws, 0, stream);
}

extern "C" int df_conv2d_wgrad_mp(df_img x, df_img dy, int ksize, int stride, int pad, float* ws, int splits,
                                  const int32_t* row_counts, int rows_per_seg, float* bias_ws, int mfma_bf16,
                                  void* stream) {
  DF_REQUIRE(!row_counts || (ksize == 1 && x.h == 1 && rows_per_seg > 0), DF_E_ARG);
  DF_REQUIRE(img_ok(x) && img_ok(dy) && ws && df_aligned16(ws), DF_E_ALIGN);
  DF_REQUIRE(x.n == dy.n && x.c % 32 == 0 && dy.c % 64 == 0, DF_E_SHAPE);
  DF_REQUIRE((ksize == 1 && stride == 1 && pad == 0) || (ksize == 3 && pad == 1 && (stride == 1 || stride == 2)), DF_E_SHAPE);
  DF_REQUIRE(dy.h == (x.h + 2 * pad - ksize) / stride + 1 && dy.w == (x.w + 2 * pad - ksize) / stride + 1, DF_E_SHAPE);
  WgradParams p;
  p.x = x; p.dy = dy; p.ws = ws; p.row_counts = row_counts; p.rows_per_seg = rows_per_seg > 0 ? rows_per_seg : 1;
  p.bias_ws = bias_ws;
  p.bf16 = mfma_bf16 != 0;
  p.stride = stride; p.pad = pad; p.K = x.c; p.N = dy.c;
  p.x_bytes = p.dy_bytes = 0;
  {
    static const int no_dma = getenv("DF_CONV_NO_DMA") ? atoi(getenv("DF_CONV_NO_DMA")) : 0;
    auto extent = [](const df_img& d) {
      return ((int64_t)(d.grp_size - 1) * d.img_stride + (int64_t)(d.n / d.grp_size - 1) * d.grp_off + (int64_t)d.h * d.w * d.ld) * 4;
    };
    const int64_t ex = extent(x), ey = extent(dy);
    if (!no_dma && x.img_stride >= 0 && dy.img_stride >= 0 && x.grp_off >= 0 && dy.grp_off >= 0 &&
        ex < (int64_t)DMA_BAD && ey < (int64_t)DMA_BAD) {
      p.x_bytes = (unsigned)ex;
      p.dy_bytes = (unsigned)ey;
    }
  }
  const int P = wgrad_chunk(ksize, stride);
  p.chunks_per_row = (dy.w + P - 1) / P;
  const int64_t chunks = (int64_t)dy.n * dy.h * p.chunks_per_row;
  DF_REQUIRE(chunks < (1ll << 31) && splits >= 1, DF_E_SHAPE);
  p.total_chunks = (int)chunks;
  p.chunks_per_split = (int)((chunks + splits - 1) / splits);
  DF_REQUIRE((int64_t)p.chunks_per_split * splits >= chunks, DF_E_SHAPE);
  dim3 grid((x.c + 63) / 64, dy.c / 64, splits);
  static const int xcd_map = getenv("DF_WGRAD_XCD") ? atoi(getenv("DF_WGRAD_XCD")) : 1;
  p.xcd_map = xcd_map;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // Measured on MI355X (bs16 step): DMA-fed kernels win for 3x3 stride 1 (register-staged 124.7 -> 4-wave DMA 128.4 ->
  // 12-wave 131-132 TFLOP/s); for 1x1 and stride 2 the register-prefetch kernels are faster (1x1: 12.0 vs 14.8 ms/step;
  // s2 needs 116 KB LDS = 1 workgroup/CU), so those keep them.  DF_WGRAD_DMA_ALL=1 forces the DMA kernels everywhere and
  // DF_WGRAD_RING=0 the 4-wave DMA form for 3x3 (A/B runs, tests).
  static const int dma_all = getenv("DF_WGRAD_DMA_ALL") ? atoi(getenv("DF_WGRAD_DMA_ALL")) : 0;
  if (p.x_bytes && ((dma_all && !(bias_ws && ksize == 1 && (dy.c % 128) == 0)) || (ksize == 3 && stride == 1) ||
                    (ksize == 3 && stride == 2 && wgrad_ring_s2()))) {
    if (wgrad_use_1x1(ksize, dy.c)) {
      const int cit = wgrad_cit(x.c);
      dim3 g1((x.c + cit - 1) / cit, dy.c / 128, splits);
      if (cit == 128) return launch_wgrad_dma(wgrad1x1_dma_kernel<128>, g1, 2 * (32 * 128 + 32 * 128) * 4, s, p);
      return launch_wgrad_dma(wgrad1x1_dma_kernel<64>, g1, 2 * (32 * 128 + 32 * 64) * 4, s, p);
    }
    auto bytes = [](int ks_, int st_) { const int xw = 31 * st_ + ks_; return (size_t)2 * (32 * 64 + ((ks_ * xw + 3) / 4) * 4 * 64) * 4; };
    if (ksize == 1) return launch_wgrad_dma(wgrad_dma_kernel<1, 1, 32>, grid, bytes(1, 1), s, p);
    const size_t ring_stage = (size_t)(32 * 64 + ((3 * 34 + 3) / 4) * 4 * 64) * 4;
    if (stride == 1 && wgrad_ring_depth(ksize, stride) == 3)
      return launch_wgrad_dma(wgrad3_ring_kernel<32, 3>, grid, 3 * ring_stage, s, p, 768);
    if (stride == 1 && wgrad_ring_depth(ksize, stride) == 2) {
      static const int bfd = getenv("DF_WGRAD_RING_BF") ? atoi(getenv("DF_WGRAD_RING_BF")) : 2;
      if (p.bf16 && bfd == 4) return launch_wgrad_dma(wgrad3_ring_kernel<32, 4, 1, true>, grid, 4 * ring_stage, s, p, 768);
      if (p.bf16 && bfd == 3) return launch_wgrad_dma(wgrad3_ring_kernel<32, 3, 1, true>, grid, 3 * ring_stage, s, p, 768);
      if (p.bf16) return launch_wgrad_dma(wgrad3_ring_kernel<32, 2, 1, true>, grid, 2 * ring_stage, s, p, 768);
      return launch_wgrad_dma(wgrad3_ring_kernel<32, 2>, grid, 2 * ring_stage, s, p, 768);
    }
    if (stride == 1) return launch_wgrad_dma(wgrad_dma_kernel<3, 1, 32>, grid, bytes(3, 1), s, p);
    if (wgrad_ring_s2()) {
      const size_t st2 = (size_t)(16 * 64 + ((3 * 33 + 3) / 4) * 4 * 64) * 4;
      if (p.bf16) return launch_wgrad_dma(wgrad3_ring_kernel<16, 2, 2, true>, grid, 2 * st2, s, p, 768);
      return launch_wgrad_dma(wgrad3_ring_kernel<16, 2, 2>, grid, 2 * st2, s, p, 768);
    }
    return launch_wgrad_dma(wgrad_dma_kernel<3, 2, 32>, grid, bytes(3, 2), s, p);
  }
  if (wgrad_use_1x1(ksize, dy.c)) {
    const int cit = wgrad_cit(x.c);
    dim3 g1((x.c + cit - 1) / cit, dy.c / 128, splits);
    if (cit == 128 && p.bf16) hipLaunchKernelGGL((wgrad1x1_kernel<128, true>), g1, dim3(256), 0, s, p);
    else if (cit == 128) hipLaunchKernelGGL((wgrad1x1_kernel<128>), g1, dim3(256), 0, s, p);
    else if (p.bf16) hipLaunchKernelGGL((wgrad1x1_kernel<64, true>), g1, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((wgrad1x1_kernel<64>), g1, dim3(256), 0, s, p);
  } else if (ksize == 1) {
    if (p.bf16) hipLaunchKernelGGL((wgrad_kernel<1, 1, 32, true>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((wgrad_kernel<1, 1, 32>), grid, dim3(256), 0, s, p);
  } else if (stride == 1) {
    if (p.bf16) hipLaunchKernelGGL((wgrad_kernel<3, 1, 32, true>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((wgrad_kernel<3, 1, 32>), grid, dim3(256), 0, s, p);
  } else {
    if (p.bf16) hipLaunchKernelGGL((wgrad_kernel<3, 2, 32, true>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((wgrad_kernel<3, 2, 32>), grid, dim3(256), 0, s, p);
  }
  DF_CHECK_LAUNCH();
  return DF_OK;
}

// fp32-accurate 3x3 stride-1 weight gradient through three bf16 planes per operand (wgrad3_x3_kernel): fp32 x and dy, splits /
// workspace / reduce as df_conv2d_wgrad_mp.  DF_WGRAD_X3=0 switches it off (df_conv2d_wgrad_x3_ok -> 0).
extern "C" int df_conv2d_wgrad_x3_ok(df_img x, df_img dy, int ksize, int stride) {
  static const int on = getenv("DF_WGRAD_X3") ? atoi(getenv("DF_WGRAD_X3")) : 1;
  auto extent = [](const df_img& d) {
    return ((int64_t)(d.grp_size - 1) * d.img_stride + (int64_t)(d.n / d.grp_size - 1) * d.grp_off + (int64_t)d.h * d.w * d.ld) * 4;
  };
  return on && ksize == 3 && stride == 1 && x.elt == 0 && dy.elt == 0 && img_ok(x) && img_ok(dy) && x.n == dy.n && x.h == dy.h &&
         x.w == dy.w && (x.c % 32) == 0 && (dy.c % 64) == 0 && (dy.w % 32) == 0 && x.img_stride >= 0 && dy.img_stride >= 0 &&
         x.grp_off >= 0 && dy.grp_off >= 0 && extent(x) < (int64_t)DMA_BAD && extent(dy) < (int64_t)DMA_BAD;
}

static int wgrad_x3_impl(df_img x, df_img dy, const float* x_amax, const float* dy_amax, int ksize, int stride, int pad, float* ws,
                         int splits, float* bias_ws, void* stream);

extern "C" int df_conv2d_wgrad_x3(df_img x, df_img dy, int ksize, int stride, int pad, float* ws, int splits, float* bias_ws,
                                  void* stream) {
  return wgrad_x3_impl(x, dy, nullptr, nullptr, ksize, stride, pad, ws, splits, bias_ws, stream);
}

// the fp16x2 form (wgrad3_x3_kernel<2>): x_amax / dy_amax = upper bounds of max|x| / max|dy| (df_absmax); shapes as the x3 form
extern "C" int df_conv2d_wgrad_h2(df_img x, df_img dy, const float* x_amax, const float* dy_amax, int ksize, int stride, int pad,
                                  float* ws, int splits, float* bias_ws, void* stream) {
  DF_REQUIRE(x_amax && dy_amax, DF_E_ARG);
  return wgrad_x3_impl(x, dy, x_amax, dy_amax, ksize, stride, pad, ws, splits, bias_ws, stream);
}

static int wgrad_x3_impl(df_img x, df_img dy, const float* x_amax, const float* dy_amax, int ksize, int stride, int pad, float* ws,
                         int splits, float* bias_ws, void* stream) {
  DF_REQUIRE(ws && df_aligned16(ws) && pad == 1 && df_conv2d_wgrad_x3_ok(x, dy, ksize, stride) == 1, DF_E_SHAPE);
  WgradParams p;
  p.x = x; p.dy = dy; p.ws = ws; p.row_counts = nullptr; p.rows_per_seg = 1; p.bias_ws = bias_ws; p.bf16 = 0;
  p.amax_x = x_amax; p.amax_dy = dy_amax;
  p.stride = 1; p.pad = 1; p.K = x.c; p.N = dy.c;
  auto extent = [](const df_img& d) {
    return ((int64_t)(d.grp_size - 1) * d.img_stride + (int64_t)(d.n / d.grp_size - 1) * d.grp_off + (int64_t)d.h * d.w * d.ld) * 4;
  };
  p.x_bytes = (unsigned)extent(x);
  p.dy_bytes = (unsigned)extent(dy);
  p.chunks_per_row = dy.w / 32;
  const int64_t chunks = (int64_t)dy.n * dy.h * p.chunks_per_row;
  DF_REQUIRE(chunks < (1ll << 31) && splits >= 1, DF_E_SHAPE);
  p.total_chunks = (int)chunks;
  p.chunks_per_split = (int)((chunks + splits - 1) / splits);
  dim3 grid((x.c + 63) / 64, dy.c / 64, splits);
  static const int xcd_map = getenv("DF_WGRAD_XCD") ? atoi(getenv("DF_WGRAD_XCD")) : 1;
  p.xcd_map = xcd_map;
  if (x_amax) return launch_wgrad_dma(wgrad3_x3_kernel<2>, grid, 2 * 2 * 17152, reinterpret_cast<hipStream_t>(stream), p, 768);
  return launch_wgrad_dma(wgrad3_x3_kernel<3>, grid, 2 * 3 * 17152, reinterpret_cast<hipStream_t>(stream), p, 768);
}

// 1x1 weight gradient, fp32 tensors, fp16x2 products (wgrad1_h2_kernel; round 5).  _ok: shapes it takes (DF_WGRAD1_H2=0 switches it
// off); _splits: split-K count (two 4-wave workgroups per CU: one resident round = 512); workspace [splits][Cout][Cin] and the reduce
// as df_conv2d_wgrad_mp.  x_amax / dy_amax = upper bounds of max|x| / max|dy| (device scalars).
static inline int wgrad1_cot(int cout) { return (cout % 128) == 0 ? 128 : 64; }
static inline int wgrad1_cit(int cin) { return cin >= 128 ? 128 : 64; }
extern "C" int df_conv2d_wgrad1_h2_ok(df_img x, df_img dy) {
  static const int on = getenv("DF_WGRAD1_H2") ? atoi(getenv("DF_WGRAD1_H2")) : 1;
  auto extent = [](const df_img& d) {
    return ((int64_t)(d.grp_size - 1) * d.img_stride + (int64_t)(d.n / d.grp_size - 1) * d.grp_off + (int64_t)d.h * d.w * d.ld) * 4;
  };
  return on && x.elt == 0 && dy.elt == 0 && img_ok(x) && img_ok(dy) && x.n == dy.n && x.h == dy.h && x.w == dy.w && (x.c % 32) == 0 &&
         (dy.c % 64) == 0 && (x.ld % 4) == 0 && (dy.ld % 4) == 0 && x.img_stride >= 0 && dy.img_stride >= 0 && x.grp_off >= 0 &&
         dy.grp_off >= 0 && extent(x) < (int64_t)DMA_BAD && extent(dy) < (int64_t)DMA_BAD;
}

extern "C" int df_conv2d_wgrad1_h2_splits(df_img x, df_img dy) {
  static const int target = getenv("DF_WGRAD1_H2_BLOCKS") ? atoi(getenv("DF_WGRAD1_H2_BLOCKS")) : 512;
  const int cot = wgrad1_cot(dy.c), cit = wgrad1_cit(x.c);
  const int tiles = ((x.c + cit - 1) / cit) * (dy.c / cot);
  const int64_t chunks = (int64_t)dy.n * dy.h * ((dy.w + 31) / 32);
  int64_t splits = (target + tiles - 1) / tiles;
  if (splits > chunks) splits = chunks;
  if (splits < 1) splits = 1;
  const int64_t cps = (chunks + splits - 1) / splits;
  return (int)((chunks + cps - 1) / cps);
}

extern "C" int df_conv2d_wgrad1_h2(df_img x, df_img dy, const float* x_amax, const float* dy_amax, float* ws, int splits, float* bias_ws,
                                   void* stream) {
  DF_REQUIRE(((x_amax && dy_amax) || (!x_amax && !dy_amax)) && ws && df_aligned16(ws), DF_E_ARG);   // both NULL: the bf16 one-plane form
  DF_REQUIRE(df_conv2d_wgrad1_h2_ok(x, dy) == 1, DF_E_SHAPE);
  WgradParams p;
  p.x = x; p.dy = dy; p.ws = ws; p.row_counts = nullptr; p.rows_per_seg = 1; p.bias_ws = bias_ws; p.bf16 = 0;
  p.amax_x = x_amax; p.amax_dy = dy_amax;
  p.stride = 1; p.pad = 0; p.K = x.c; p.N = dy.c;
  auto extent = [](const df_img& d) {
    return ((int64_t)(d.grp_size - 1) * d.img_stride + (int64_t)(d.n / d.grp_size - 1) * d.grp_off + (int64_t)d.h * d.w * d.ld) * 4;
  };
  p.x_bytes = (unsigned)extent(x);
  p.dy_bytes = (unsigned)extent(dy);
  p.chunks_per_row = (dy.w + 31) / 32;
  const int64_t chunks = (int64_t)dy.n * dy.h * p.chunks_per_row;
  DF_REQUIRE(chunks < (1ll << 31) && splits >= 1, DF_E_SHAPE);
  p.total_chunks = (int)chunks;
  p.chunks_per_split = (int)((chunks + splits - 1) / splits);
  const int cot = wgrad1_cot(dy.c), cit = wgrad1_cit(x.c);
  dim3 grid((x.c + cit - 1) / cit, dy.c / cot, splits);
  static const int xcd_map = getenv("DF_WGRAD_XCD") ? atoi(getenv("DF_WGRAD_XCD")) : 1;
  p.xcd_map = xcd_map;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t lds = (size_t)2 * 2 * (cot + cit) * 64;      // two stages x two planes x (COT + CIT) / 32 halves x 32 px x 64 B
  if (!x_amax) {
    if (cot == 128 && cit == 128) return launch_wgrad_dma(wgrad1_h2_kernel<128, 128, true>, grid, lds / 2, s, p);
    if (cot == 128) return launch_wgrad_dma(wgrad1_h2_kernel<128, 64, true>, grid, lds / 2, s, p);
    if (cit == 128) return launch_wgrad_dma(wgrad1_h2_kernel<64, 128, true>, grid, lds / 2, s, p);
    return launch_wgrad_dma(wgrad1_h2_kernel<64, 64, true>, grid, lds / 2, s, p);
  }
  if (cot == 128 && cit == 128) return launch_wgrad_dma(wgrad1_h2_kernel<128, 128>, grid, lds, s, p);
  if (cot == 128) return launch_wgrad_dma(wgrad1_h2_kernel<128, 64>, grid, lds, s, p);
  if (cit == 128) return launch_wgrad_dma(wgrad1_h2_kernel<64, 128>, grid, lds, s, p);
  return launch_wgrad_dma(wgrad1_h2_kernel<64, 64>, grid, lds, s, p);
}

// 3x3 stride-2 weight gradient, fp32 tensors, fp16x2 products (wgrad3s2_h2_kernel; round 5).  Geometry as df_conv2d_wgrad_mp's
// stride-2 case (pad 1; Hout = (H - 1) / 2 + 1); DF_WGRAD_S2_H2=0 switches it off.  Workspace [splits][Cout][9][Cin].
extern "C" int df_conv2d_wgrad_s2_h2_ok(df_img x, df_img dy) {
  static const int on = getenv("DF_WGRAD_S2_H2") ? atoi(getenv("DF_WGRAD_S2_H2")) : 1;
  auto extent = [](const df_img& d) {
    return ((int64_t)(d.grp_size - 1) * d.img_stride + (int64_t)(d.n / d.grp_size - 1) * d.grp_off + (int64_t)d.h * d.w * d.ld) * 4;
  };
  return on && x.elt == 0 && dy.elt == 0 && img_ok(x) && img_ok(dy) && x.n == dy.n && dy.h == (x.h - 1) / 2 + 1 && dy.w == (x.w - 1) / 2 + 1 &&
         (x.c % 32) == 0 && (dy.c % 64) == 0 && (x.ld % 4) == 0 && (dy.ld % 4) == 0 && x.img_stride >= 0 && dy.img_stride >= 0 &&
         x.grp_off >= 0 && dy.grp_off >= 0 && extent(x) < (int64_t)DMA_BAD && extent(dy) < (int64_t)DMA_BAD;
}

extern "C" int df_conv2d_wgrad_s2_h2_splits(df_img x, df_img dy) {
  static const int target = getenv("DF_WGRAD_S2_H2_BLOCKS") ? atoi(getenv("DF_WGRAD_S2_H2_BLOCKS")) : 256;   // one 12-wave workgroup per CU
  const int tiles = ((x.c + 63) / 64) * (dy.c / 64);
  const int64_t chunks = (int64_t)dy.n * dy.h * ((dy.w + 15) / 16);
  int64_t splits = (target + tiles - 1) / tiles;
  if (splits > chunks) splits = chunks;
  if (splits < 1) splits = 1;
  const int64_t cps = (chunks + splits - 1) / splits;
  return (int)((chunks + cps - 1) / cps);
}

extern "C" int df_conv2d_wgrad_s2_h2(df_img x, df_img dy, const float* x_amax, const float* dy_amax, float* ws, int splits,
                                     float* bias_ws, void* stream) {
  DF_REQUIRE(((x_amax && dy_amax) || (!x_amax && !dy_amax)) && ws && df_aligned16(ws), DF_E_ARG);   // both NULL: the bf16 one-plane form
  DF_REQUIRE(df_conv2d_wgrad_s2_h2_ok(x, dy) == 1, DF_E_SHAPE);
  WgradParams p;
  p.x = x; p.dy = dy; p.ws = ws; p.row_counts = nullptr; p.rows_per_seg = 1; p.bias_ws = bias_ws; p.bf16 = 0;
  p.amax_x = x_amax; p.amax_dy = dy_amax;
  p.stride = 2; p.pad = 1; p.K = x.c; p.N = dy.c;
  auto extent = [](const df_img& d) {
    return ((int64_t)(d.grp_size - 1) * d.img_stride + (int64_t)(d.n / d.grp_size - 1) * d.grp_off + (int64_t)d.h * d.w * d.ld) * 4;
  };
  p.x_bytes = (unsigned)extent(x);
  p.dy_bytes = (unsigned)extent(dy);
  p.chunks_per_row = (dy.w + 15) / 16;
  const int64_t chunks = (int64_t)dy.n * dy.h * p.chunks_per_row;
  DF_REQUIRE(chunks < (1ll << 31) && splits >= 1, DF_E_SHAPE);
  p.total_chunks = (int)chunks;
  p.chunks_per_split = (int)((chunks + splits - 1) / splits);
  dim3 grid((x.c + 63) / 64, dy.c / 64, splits);
  static const int xcd_map = getenv("DF_WGRAD_XCD") ? atoi(getenv("DF_WGRAD_XCD")) : 1;
  p.xcd_map = xcd_map;
  if (!x_amax) return launch_wgrad_dma(wgrad3s2_h2_kernel<true>, grid, 2 * 15104, reinterpret_cast<hipStream_t>(stream), p, 768);
  return launch_wgrad_dma(wgrad3s2_h2_kernel<false>, grid, 2 * 2 * 15104, reinterpret_cast<hipStream_t>(stream), p, 768);
}

// PRE-SPLIT fp16x2 tensors (round 4): x and dy are h2 images (df_img.elt = 2: per pixel and 32-channel chunk one 128-byte line
// [32 fp16 hi | 32 fp16 lo] of the value scaled by df_h2_scale(bound)); x_bound / dy_bound = the device scalars that DEFINED those
// scales (the producers' bounds).  Shapes as df_conv2d_wgrad_x3 (3x3, stride 1, W % 32 == 0); splits / workspace / reduce as
// df_conv2d_wgrad_mp.  DF_WGRAD_H2P_DEPTH = ring depth (2..4, default 4).
extern "C" int df_conv2d_wgrad_h2p_ok(df_img x, df_img dy, int ksize, int stride) {
  if (x.elt != 2 || dy.elt != 2) return 0;
  df_img xf = x, yf = dy;
  xf.elt = yf.elt = 0;                    // (geometry and extents are those of the fp32 tensor: 4 bytes per element)
  return df_conv2d_wgrad_x3_ok(xf, yf, ksize, stride) == 1 && (x.ld % 32) == 0 && (dy.ld % 32) == 0 && (x.img_stride % 32) == 0 &&
         (dy.img_stride % 32) == 0 && (x.grp_off % 32) == 0 && (dy.grp_off % 32) == 0 && (((uintptr_t)x.ptr | (uintptr_t)dy.ptr) & 127) == 0;
}

// split-K count for df_conv2d_wgrad_h2p: its 144 KB ring leaves ONE 12-wave workgroup per CU, so one resident round is 256
// workgroups (DF_WGRAD_H2P_BLOCKS); every split non-empty
extern "C" int df_conv2d_wgrad_h2p_splits(df_img x, df_img dy) {
  static const int target = getenv("DF_WGRAD_H2P_BLOCKS") ? atoi(getenv("DF_WGRAD_H2P_BLOCKS")) : 256;
  const int tiles = ((x.c + 63) / 64) * (dy.c / 64);
  const int64_t chunks = (int64_t)dy.n * dy.h * ((dy.w + 31) / 32);
  int64_t splits = (target + tiles - 1) / tiles;
  if (splits > chunks) splits = chunks;
  if (splits < 1) splits = 1;
  const int64_t cps = (chunks + splits - 1) / splits;
  return (int)((chunks + cps - 1) / cps);
}

extern "C" int df_conv2d_wgrad_h2p(df_img x, df_img dy, const float* x_bound, const float* dy_bound, int ksize, int stride, int pad,
                                   float* ws, int splits, float* bias_ws, void* stream) {
  DF_REQUIRE(x_bound && dy_bound && ws && df_aligned16(ws), DF_E_ARG);
  DF_REQUIRE(pad == 1 && df_conv2d_wgrad_h2p_ok(x, dy, ksize, stride) == 1, DF_E_SHAPE);
  WgradParams p;
  p.x = x; p.dy = dy; p.ws = ws; p.row_counts = nullptr; p.rows_per_seg = 1; p.bias_ws = bias_ws; p.bf16 = 0;
  p.amax_x = x_bound; p.amax_dy = dy_bound;
  p.stride = 1; p.pad = 1; p.K = x.c; p.N = dy.c;
  auto extent = [](const df_img& d) {
    return ((int64_t)(d.grp_size - 1) * d.img_stride + (int64_t)(d.n / d.grp_size - 1) * d.grp_off + (int64_t)d.h * d.w * d.ld) * 4;
  };
  p.x_bytes = (unsigned)extent(x);
  p.dy_bytes = (unsigned)extent(dy);
  p.chunks_per_row = dy.w / 32;
  const int64_t chunks = (int64_t)dy.n * dy.h * p.chunks_per_row;
  DF_REQUIRE(chunks < (1ll << 31) && splits >= 1, DF_E_SHAPE);
  p.total_chunks = (int)chunks;
  p.chunks_per_split = (int)((chunks + splits - 1) / splits);
  dim3 grid((x.c + 63) / 64, dy.c / 64, splits);
  static const int xcd_map = getenv("DF_WGRAD_XCD") ? atoi(getenv("DF_WGRAD_XCD")) : 1;
  p.xcd_map = xcd_map;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  static const int depth = getenv("DF_WGRAD_H2P_DEPTH") ? atoi(getenv("DF_WGRAD_H2P_DEPTH")) : 4;
  const size_t stage = 2 * 17408;
  if (depth == 2) return launch_wgrad_dma(wgrad3_h2p_kernel<2>, grid, 2 * stage, s, p, 768);
  if (depth == 3) return launch_wgrad_dma(wgrad3_h2p_kernel<3>, grid, 3 * stage, s, p, 768);
  static const int pipe = getenv("DF_WGRAD_H2P_PIPE") ? atoi(getenv("DF_WGRAD_H2P_PIPE")) : 1;   // 0: the round-4 loop, for A/B
  if (pipe) return launch_wgrad_dma(wgrad3_h2p_kernel<4, true>, grid, 4 * stage + 6 * 768 * 4, s, p, 768);
  return launch_wgrad_dma(wgrad3_h2p_kernel<4>, grid, 4 * stage, s, p, 768);
}

// bf16-STORAGE training: 3x3 stride-1 weight gradient of bfloat16 x and dy (wgrad3_tr_kernel); splits / workspace / reduce as
// df_conv2d_wgrad_mp (df_conv2d_wgrad_splits with the same shapes).
extern "C" int df_conv2d_wgrad_bf16(df_img x, df_img dy, int ksize, int stride, int pad, float* ws, int splits, float* bias_ws,
                                    void* stream) {
  DF_REQUIRE(img_ok(x, true) && img_ok(dy, true) && x.elt == 1 && dy.elt == 1 && ws && df_aligned16(ws), DF_E_ALIGN);
  DF_REQUIRE(ksize == 3 && stride == 1 && pad == 1, DF_E_SHAPE);
  DF_REQUIRE(x.n == dy.n && x.h == dy.h && x.w == dy.w && x.c % 32 == 0 && dy.c % 64 == 0 && (dy.w % 32) == 0, DF_E_SHAPE);
  WgradParams p;
  p.x = x; p.dy = dy; p.ws = ws; p.row_counts = nullptr; p.rows_per_seg = 1; p.bias_ws = bias_ws; p.bf16 = 1;
  p.stride = 1; p.pad = 1; p.K = x.c; p.N = dy.c;
  auto extent = [](const df_img& d) {
    return ((int64_t)(d.grp_size - 1) * d.img_stride + (int64_t)(d.n / d.grp_size - 1) * d.grp_off + (int64_t)d.h * d.w * d.ld) * 2;
  };
  const int64_t ex = extent(x), ey = extent(dy);
  DF_REQUIRE(x.img_stride >= 0 && dy.img_stride >= 0 && x.grp_off >= 0 && dy.grp_off >= 0 && ex < (int64_t)DMA_BAD && ey < (int64_t)DMA_BAD,
             DF_E_SHAPE);
  p.x_bytes = (unsigned)ex;
  p.dy_bytes = (unsigned)ey;
  p.chunks_per_row = dy.w / 32;
  const int64_t chunks = (int64_t)dy.n * dy.h * p.chunks_per_row;
  DF_REQUIRE(chunks < (1ll << 31) && splits >= 1, DF_E_SHAPE);
  p.total_chunks = (int)chunks;
  p.chunks_per_split = (int)((chunks + splits - 1) / splits);
  dim3 grid((x.c + 63) / 64, dy.c / 64, splits);
  static const int xcd_map = getenv("DF_WGRAD_XCD") ? atoi(getenv("DF_WGRAD_XCD")) : 1;
  p.xcd_map = xcd_map;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  static const int depth = getenv("DF_WGRAD_TR_DEPTH") ? atoi(getenv("DF_WGRAD_TR_DEPTH")) : 4;
  const size_t stage = 4096 + 13 * 1024;
  if (depth == 2) return launch_wgrad_dma(wgrad3_tr_kernel<2>, grid, 2 * stage, s, p, 768);
  if (depth == 3) return launch_wgrad_dma(wgrad3_tr_kernel<3>, grid, 3 * stage, s, p, 768);
  return launch_wgrad_dma(wgrad3_tr_kernel<4>, grid, 4 * stage, s, p, 768);
}

extern "C" int df_conv2d_wgrad_reduce(const float* ws, int splits, int cout, int taps, int cin, float* dw,
                                      int64_t ld_co, int accumulate, void* stream) {
  DF_REQUIRE(ws && dw && splits >= 1, DF_E_ARG);
  const int64_t per = (int64_t)cout * taps * cin;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((per + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), ws, splits, per, taps * cin, dw, ld_co, accumulate);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

extern "C" int df_conv2d_wgrad_reduce_bias(const float* ws, int splits, int cout, int taps, int cin, float* dw, int64_t ld_co,
                                           int accumulate, const float* bias_ws, float* db, void* stream) {
  DF_REQUIRE(ws && dw && bias_ws && db && splits >= 1 && cout >= 1, DF_E_ARG);
  const int64_t per = (int64_t)cout * taps * cin;
  const int nblk_w = (int)((per + 255) / 256);
  hipLaunchKernelGGL(wgrad_reduce_bias_kernel, dim3((unsigned)(nblk_w + (cout + 31) / 32)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), ws, splits, per, taps * cin, dw, ld_co, accumulate, nblk_w, bias_ws, cout, db);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

extern "C" int df_weight_transpose(const float* w, float* wt, int cout, int taps, int cin, void* stream) {
  DF_REQUIRE(w && wt, DF_E_ARG);
  const int64_t total = (int64_t)cout * taps * cin;
  hipLaunchKernelGGL(weight_transpose_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), w, wt, cout, taps, cin);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

