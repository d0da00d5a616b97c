// Pillar feature net, BACKWARD side, and the sparse-edge kernels of the UNet.  The forward pillariser (voxelise, sort, feature
// net, canvas) is csrc/pillar_bands.hip; the first-generation forward kernels that lived here (keys / scan / compact / library
// radix sort / gather / cells / stats / canvas) were retired in round 3 together with the rocPRIM dependency.  What is here:
//   pfn_bn_finalize / running     per-sample BatchNorm1d statistics of the feature net from the band kernel's partials
//   pfn_bwd_stats / finalize / weights   backward of Linear(9->32) + BN1d + ReLU + per-pillar mean|max over the sorted runs
//   pillar_input_grad, sparse_in_wgrad, sparse_conv3x3, sparse_wgrad3x3   the UNet's first / last convolutions evaluated at
//                                 occupied pillars only (the canvas is >90 % zeros; the decoder reads pc0's cells only)
//   cell sort                     counting sort of caller-supplied cell keys for the stand-alone decoder head (pack_infos)
// ([REF deflow.py:27-30,82-83]; the mmcv / OpenSceneFlow sources are absent -- the algorithm is restated in oracle/ref_torch.py.)
// No float atomics anywhere: every sum has one fixed order.
#include <cstdlib>
#include <cstring>

#include "common.h"
#include "pillar_common.h"

namespace {

constexpr int CELLS_PER_BLOCK = 32;

// Sparse iteration: the feature-net kernels visit OCCUPIED pillars only.  The sorted key array is globally ordered by
// b * H * W + cell, so sample b owns sorted positions [off, off + cnt) with off = sum of counts[0..b); a position whose
// key differs from its predecessor's is a pillar head, and its run is [i, cell_rng[key].end).
struct SampleRange {
  int off, cnt;
};
__device__ __forceinline__ SampleRange sample_range(const int32_t* __restrict__ counts, int b) {
  SampleRange r;
  r.off = 0;
  for (int k = 0; k < b; ++k) r.off += counts[k];
  r.cnt = counts[b];
  return r;
}

// helpers of the 16-bit-MFMA forms of the sparse kernels (round 6): the fp16x2 scale of conv_common.h, and the 16-byte-slot swizzle
// that makes one ds_read_b128 per weight fragment conflict-free (gemm_dma.h: g = (0, 2, 3, 1) over (row >> 2) & 3)
__device__ __forceinline__ float sp_h2_scale(float amax) {
  const int e = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 255u);
  const int f = min(max(268 - e, 1), 254);
  return __builtin_bit_cast(float, (unsigned)f << 23);
}
__device__ __forceinline__ int sp_g4(int x) { return (0x78 >> (2 * x)) & 3; }

// reduce NV floats per lane across the 32 cell groups of a block (lanes with equal `sub`)
template <int NV>
__device__ __forceinline__ void reduce_groups(float (&v)[NV], float* lds /*[256*NV]*/) {
  const int sub = threadIdx.x & 7;
#pragma unroll
  for (int k = 0; k < NV; ++k) lds[k * 256 + threadIdx.x] = v[k];
  __syncthreads();
  if (threadIdx.x < 8) {
    for (int gi = 1; gi < CELLS_PER_BLOCK; ++gi)
#pragma unroll
      for (int k = 0; k < NV; ++k) v[k] += lds[k * 256 + gi * 8 + sub];
  }
}

// one workgroup per sample (the feature net is called once per sample: BatchNorm1d statistics are per sample): fp64 sums
// of the per-band / per-block partials -> bn_ss[b] = (scale, shift, mean, invstd); (mean, unbiased var) of the sample is
// left in the first partial slot for the running-statistics pass below, which must run in sample order
// canvas_bound (round 4, optional): an a-priori bound of max |canvas| = max over (sample, channel) of |scale| U_c + |shift| with
// U_c = sum_j |W[c][j]| fmax_j >= |linear output|: the nine point features are bounded by the geometry (coordinates by the
// range, offsets to the pillar mean by the voxel size, offsets to the centre by half of it).  It spares the fp16x2 consumers of
// the canvas (the decoder's 1x1 skip conv writes a pre-split concatenation) one pass over the 1 GB canvas.  Integer atomic max.
__global__ __launch_bounds__(1024) void pfn_bn_finalize_kernel(float* __restrict__ partial, int nblk,
                                                               const int32_t* __restrict__ counts, const float* gamma,
                                                               const float* beta, float eps, float* __restrict__ bn_ss,
                                                               const float* __restrict__ w_pfn, df_pillar_geom g,
                                                               unsigned* __restrict__ canvas_bound) {
  __shared__ double red[2][32][32];
  const int c = threadIdx.x & 31, tl = threadIdx.x >> 5;  // 32 channels x 32 partial lanes
  const int b = blockIdx.x;
  float* o = bn_ss + (int64_t)b * 128;
  const int cnt = counts[b];
  double s1 = 0.0, s2 = 0.0;
  for (int k = tl; k < nblk; k += 32) {
    const float* q = partial + (((int64_t)b * nblk + k) * 32 + c) * 2;
    s1 += (double)q[0];
    s2 += (double)q[1];
  }
  red[0][tl][c] = s1;
  red[1][tl][c] = s2;
  __syncthreads();
  if (tl != 0) return;
  for (int k = 1; k < 32; ++k) {
    s1 += red[0][k][c];
    s2 += red[1][k][c];
  }
  float* mv = partial + (((int64_t)b * nblk) * 32 + c) * 2;
  if (cnt <= 0) {
    o[c] = 0.f; o[32 + c] = 0.f; o[64 + c] = 0.f; o[96 + c] = 0.f;
    mv[0] = 0.f; mv[1] = 0.f;
    return;
  }
  const double mean = s1 / cnt;
  double var = s2 / cnt - mean * mean;
  if (var < 0.0) var = 0.0;
  const double invstd = 1.0 / sqrt(var + (double)eps);
  const double ga = gamma ? (double)gamma[c] : 1.0, be = beta ? (double)beta[c] : 0.0;
  o[c] = (float)(ga * invstd);
  o[32 + c] = (float)(be - mean * ga * invstd);
  o[64 + c] = (float)mean;
  o[96 + c] = (float)invstd;
  if (canvas_bound) {
    const float fx = fmaxf(fabsf(g.minx), fabsf(g.minx + g.gx * g.vx)), fy = fmaxf(fabsf(g.miny), fabsf(g.miny + g.gy * g.vy)),
                fz = fmaxf(fabsf(g.minz), fabsf(g.minz + g.gz * g.vz));
    const float fm[9] = {fx, fy, fz, g.vx, g.vy, g.vz, 0.5f * g.vx, 0.5f * g.vy, 0.5f * g.vz};
    float U = 0.f;
    for (int j = 0; j < 9; ++j) U += fabsf(w_pfn[c * 9 + j]) * fm[j];
    const unsigned bb = __builtin_bit_cast(unsigned, (fabsf(o[c]) * U + fabsf(o[32 + c])) * 1.001f);
    if (bb > __atomic_load_n(canvas_bound, __ATOMIC_RELAXED)) atomicMax(canvas_bound, bb);
  }
  mv[0] = (float)mean;
  mv[1] = (float)(cnt > 1 ? var * cnt / (cnt - 1.0) : var);
}

// running statistics: B successive momentum updates (one module call per sample), skipped for calls with < 2 points
__global__ void pfn_bn_running_kernel(const float* __restrict__ partial, int B, int nblk, const int32_t* __restrict__ counts,
                                      float momentum, float* running_mean, float* running_var) {
  const int c = threadIdx.x;
  double rm = (double)running_mean[c], rv = (double)running_var[c];
  for (int b = 0; b < B; ++b) {
    if (counts[b] <= 1) continue;
    const float* mv = partial + (((int64_t)b * nblk) * 32 + c) * 2;
    rm = (float)((1.0 - momentum) * rm + momentum * (double)mv[0]);
    rv = (float)((1.0 - momentum) * rv + momentum * (double)mv[1]);
  }
  running_mean[c] = (float)rm;
  running_var[c] = (float)rv;
}

// backward pass A: per-sample sums of (g_hat, g_hat * xhat) where g_hat = dL/d(BN output) after the ReLU mask
template <int MODE>
__global__ __launch_bounds__(256) void pfn_bwd_stats_kernel(const float* __restrict__ pts,
                                                            const int32_t* __restrict__ cell_rng,
    const uint32_t* __restrict__ key_sorted, const int32_t* __restrict__ counts, df_pillar_geom g,
                                                            const float* __restrict__ w_pfn,
                                                            const float* __restrict__ bn_ss, int bn_sample_stride,
                                                            df_img gout, float* __restrict__ partial) {
  __shared__ float lds[256 * 8];
  const int b = blockIdx.y, sub = threadIdx.x & 7, grp = threadIdx.x >> 3;
  const int ncell = g.gx * g.gy;
  PfnCtx c;
  pfn_load_w(c, w_pfn, sub);
  pfn_load_bn(c, bn_ss + (int64_t)b * bn_sample_stride, sub);
  const float* __restrict__ gp = reinterpret_cast<const float*>(gout.ptr) + df_img_base(gout, b);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const SampleRange sr = sample_range(counts, b);
  for (int i0 = sr.off + blockIdx.x * CELLS_PER_BLOCK + grp; i0 < sr.off + sr.cnt; i0 += gridDim.x * CELLS_PER_BLOCK) {
    const uint32_t key = key_sorted[i0];
    if (i0 > sr.off && key_sorted[i0 - 1] == key) continue;  // not a pillar head
    const int cell = (int)(key - (uint32_t)b * (uint32_t)ncell);
    const int s = i0, e = cell_rng[2 * (int64_t)key + 1];
    if (e <= s) continue;
    float mx, my, mz, ctx, cty, ctz;
    pfn_mean(pts, s, e, mx, my, mz);
    pfn_centre(g, cell, ctx, cty, ctz);
    const f32x4 gc = ld4(gp + (int64_t)cell * gout.ld + 4 * sub);
    const float inv = MODE == 0 ? 1.f / (float)(e - s) : 1.f;
    int am[4] = {0, 0, 0, 0};
    if (MODE == 1) pfn_argmax(c, pts, s, e, mx, my, mz, ctx, cty, ctz, am);
    for (int i = s; i < e; ++i) {
      float f[9], u[4];
      pfn_feat(pts + (int64_t)i * 3, mx, my, mz, ctx, cty, ctz, f);
      pfn_linear(c, f, u);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float yh = fmaf(u[k], c.sc[k], c.sh[k]);
        const float gh = (yh > 0.f && (MODE == 0 || i == am[k])) ? gc[k] * inv : 0.f;
        acc[k] += gh;
        acc[4 + k] += gh * ((u[k] - c.mu[k]) * c.is[k]);
      }
    }
  }
  reduce_groups<8>(acc, lds);
  if (threadIdx.x < 8) {
    float* o = partial + (((int64_t)b * gridDim.x + blockIdx.x) * 32 + 4 * sub) * 2;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      o[k * 2 + 0] = acc[k];
      o[k * 2 + 1] = acc[4 + k];
    }
  }
}

__global__ void pfn_bwd_finalize_kernel(const float* __restrict__ partial, int B, int nblk,
                                        const int32_t* __restrict__ counts, float* dgamma, float* dbeta,
                                        int accumulate, float* __restrict__ coef) {
  __shared__ double red[2][32][32];
  const int c = threadIdx.x & 31, tl = threadIdx.x >> 5;
  double tg = 0.0, tb = 0.0;
  for (int b = 0; b < B; ++b) {
    double s1 = 0.0, s2 = 0.0;
    for (int k = tl; k < nblk; k += 32) {
      const float* q = partial + (((int64_t)b * nblk + k) * 32 + c) * 2;
      s1 += (double)q[0];
      s2 += (double)q[1];
    }
    red[0][tl][c] = s1;
    red[1][tl][c] = s2;
    __syncthreads();
    if (tl == 0)
      for (int k = 1; k < 32; ++k) {
        s1 += red[0][k][c];
        s2 += red[1][k][c];
      }
    __syncthreads();
    if (tl != 0) continue;
    const int cnt = counts[b];
    coef[((int64_t)b * 2 + 0) * 32 + c] = cnt > 0 ? (float)(s1 / cnt) : 0.f;
    coef[((int64_t)b * 2 + 1) * 32 + c] = cnt > 0 ? (float)(s2 / cnt) : 0.f;
    tb += s1;
    tg += s2;
  }
  if (tl != 0) return;
  dgamma[c] = accumulate ? (float)((double)dgamma[c] + tg) : (float)tg;
  dbeta[c] = accumulate ? (float)((double)dbeta[c] + tb) : (float)tb;
}

// backward pass B: dW[32][9] partial sums of du (x) f
template <int MODE>
__global__ __launch_bounds__(256) void pfn_bwd_weights_kernel(const float* __restrict__ pts,
                                                              const int32_t* __restrict__ cell_rng,
    const uint32_t* __restrict__ key_sorted, const int32_t* __restrict__ counts, df_pillar_geom g,
                                                              const float* __restrict__ w_pfn,
                                                              const float* __restrict__ bn_ss, int bn_sample_stride,
                                                              const float* __restrict__ coef, df_img gout,
                                                              float* __restrict__ dw_partial) {
  __shared__ float lds[256 * 36];
  const int b = blockIdx.y, sub = threadIdx.x & 7, grp = threadIdx.x >> 3;
  const int ncell = g.gx * g.gy;
  PfnCtx c;
  pfn_load_w(c, w_pfn, sub);
  pfn_load_bn(c, bn_ss + (int64_t)b * bn_sample_stride, sub);
  float c1[4], c2[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    c1[k] = coef[((int64_t)b * 2 + 0) * 32 + 4 * sub + k];
    c2[k] = coef[((int64_t)b * 2 + 1) * 32 + 4 * sub + k];
  }
  const float* __restrict__ gp = reinterpret_cast<const float*>(gout.ptr) + df_img_base(gout, b);
  float acc[36];
#pragma unroll
  for (int k = 0; k < 36; ++k) acc[k] = 0.f;
  const SampleRange sr = sample_range(counts, b);
  for (int i0 = sr.off + blockIdx.x * CELLS_PER_BLOCK + grp; i0 < sr.off + sr.cnt; i0 += gridDim.x * CELLS_PER_BLOCK) {
    const uint32_t key = key_sorted[i0];
    if (i0 > sr.off && key_sorted[i0 - 1] == key) continue;  // not a pillar head
    const int cell = (int)(key - (uint32_t)b * (uint32_t)ncell);
    const int s = i0, e = cell_rng[2 * (int64_t)key + 1];
    if (e <= s) continue;
    float mx, my, mz, ctx, cty, ctz;
    pfn_mean(pts, s, e, mx, my, mz);
    pfn_centre(g, cell, ctx, cty, ctz);
    const f32x4 gc = ld4(gp + (int64_t)cell * gout.ld + 4 * sub);
    const float inv = MODE == 0 ? 1.f / (float)(e - s) : 1.f;
    int am[4] = {0, 0, 0, 0};
    if (MODE == 1) pfn_argmax(c, pts, s, e, mx, my, mz, ctx, cty, ctz, am);
    for (int i = s; i < e; ++i) {
      float f[9], u[4];
      pfn_feat(pts + (int64_t)i * 3, mx, my, mz, ctx, cty, ctz, f);
      pfn_linear(c, f, u);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float yh = fmaf(u[k], c.sc[k], c.sh[k]);
        const float gh = (yh > 0.f && (MODE == 0 || i == am[k])) ? gc[k] * inv : 0.f;
        const float xh = (u[k] - c.mu[k]) * c.is[k];
        const float du = c.sc[k] * (gh - c1[k] - xh * c2[k]);
#pragma unroll
        for (int j = 0; j < 9; ++j) acc[k * 9 + j] = fmaf(du, f[j], acc[k * 9 + j]);
      }
    }
  }
  reduce_groups<36>(acc, lds);
  if (threadIdx.x < 8) {
    float* o = dw_partial + ((int64_t)b * gridDim.x + blockIdx.x) * 288 + 4 * sub * 9;
#pragma unroll
    for (int k = 0; k < 36; ++k) o[k] = acc[k];
  }
}

bool geom_ok(const df_pillar_geom& g) {
  return g.gx > 0 && g.gy > 0 && g.gz == 1 && g.vx > 0.f && g.vy > 0.f && g.vz > 0.f;
}


// ---------------------------------------------------------------------------------------- sparse canvas gradient ---
// d(canvas) is only ever read at OCCUPIED pillars (the canvas is a scatter of pillar features; empty cells are constants
// with nothing upstream), yet dense kernels used to produce it for all H*W cells: the stride-2 data gradient of the first
// encoder conv (32 <- 64 channels) and the data gradient of the decoder's 1x1 skip conv on the canvas -- 3.1 ms per step
// for a tensor that is 80-90 % dead.  This kernel evaluates both only at the occupied cells of one cloud and adds them
// to what is already there (the decoder's gather backward, which is a cheap dense stream):
//   dcanvas[b,y,x,c] += sum_{ky,kx of (y,x)'s parity} sum_co dy1[g*B+b, (y+1-ky)/2, (x+1-kx)/2, co] w1[co,ky,kx,c]  (conv, s2, pad 1)
//                     + sum_j dskip[b,y,x,j] w3[j, 32 g + c]                                                      (1x1 skip conv)
// as small MFMA GEMMs: a wave scans 64 consecutive sorted points, ballots the pillar heads, and for each of the four
// output-parity classes (which fix the reachable taps: 1, 2, 2 or 4 of 9) multiplies batches of up to 16 cells
// [16 x (taps + 1) * 64] by the weight rows [(taps + 1) * 64 x 32] with v_mfma_f32_16x16x4_f32.  A-operand rows come
// straight from global memory (lane (cell, q) holds 16 consecutive channels of its cell's gradient row: the MFMA k index
// is the pair (q, step)), the weights of both convs sit in LDS (82 KB, one 16-wave workgroup per CU).  The first version
// did the same sums as per-lane FMAs with one LDS weight read each and was LDS-bound at 1.0 ms per cloud.
// Cells nobody reads keep whatever the dense producers left there.
constexpr int PG_THREADS = 1024;
struct PillarGradParams {
  const uint32_t* key_sorted;
  const int32_t* counts;
  int B, H, W, cloud, accumulate;
  const float* dy1;
  const float* w1;
  df_img dskip;
  const float* w3;
  df_img dcanvas;
};

__global__ __launch_bounds__(PG_THREADS) void pillar_input_grad_kernel(PillarGradParams p) {
#if defined(__HIP_DEVICE_COMPILE__)  // buffer-resource builtins only exist in the device pass
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* W1s = lds;                       // [9][64][32]
  float* W3s = W1s + 9 * 64 * 32;         // [64][32]
  int* RowCell = reinterpret_cast<int*>(W3s + 64 * 32);   // [16 waves][16]
  const int tid = threadIdx.x;
  const int b = blockIdx.y, g = p.cloud;
  {  // all loads first (18 + 2 per thread), then the LDS writes: one round trip instead of twenty
    float wv[18], w3v[2];
#pragma unroll
    for (int k = 0; k < 18; ++k) {
      const int i = tid + PG_THREADS * k;
      wv[k] = p.w1[(((i >> 5) & 63) * 9 + (i >> 11)) * 32 + (i & 31)];
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = tid + PG_THREADS * k;
      w3v[k] = p.w3[(i >> 5) * 64 + 32 * g + (i & 31)];
    }
#pragma unroll
    for (int k = 0; k < 18; ++k) W1s[tid + PG_THREADS * k] = wv[k];
#pragma unroll
    for (int k = 0; k < 2; ++k) W3s[tid + PG_THREADS * k] = w3v[k];
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6, li = lane & 15, lq = lane >> 4;
  int* rowcell = RowCell + wave * 16;
  const int ncell = p.H * p.W, h2 = p.H >> 1, w2 = p.W >> 1;
  const float* dy1 = p.dy1 + (int64_t)(g * p.B + b) * h2 * w2 * 64;
  const float* dsk = reinterpret_cast<const float*>(p.dskip.ptr) + df_img_base(p.dskip, b);
  float* out = reinterpret_cast<float*>(p.dcanvas.ptr) + df_img_base(p.dcanvas, b);
  const SampleRange sr = sample_range(p.counts, b);
  const int end = sr.off + sr.cnt;
  auto lds_fence = [&]() {   // a wave's own LDS writes are visible to its reads in program order; stop compiler reordering
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  // acc[nt] += A_row(li)[16 lq .. 16 lq + 15] x Ws[(16 lq + s) * 32 + 16 nt + li], s = 0..15
  auto mma_row = [&](const f32x4 (&a4)[4], const float* Ws, f32x4 (&acc)[2]) {
    const float* wl = Ws + 16 * lq * 32 + li;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float av = a4[s >> 2][s & 3];
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wl[s * 32], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wl[s * 32 + 16], acc[1], 0, 0, 0);
    }
  };
  // (round 4, second session) the rows of a batch -- the skip gradient and the one to four reachable taps of dy1 -- are fetched
  // TOGETHER with branch-free buffer loads (a missing row's offset is out of range and reads 0) and multiplied as they land; before,
  // each row was a load round trip in front of its 32 MFMAs
  constexpr unsigned OOB = 0xF0000000u;
  const __amdgpu_buffer_rsrc_t dskr = __builtin_amdgcn_make_buffer_rsrc((void*)dsk, 0, (unsigned)(((int64_t)(ncell - 1) * p.dskip.ld + 64) * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t dyr = __builtin_amdgcn_make_buffer_rsrc((void*)dy1, 0, (unsigned)((int64_t)h2 * w2 * 64 * 4), 0x00020000);
  auto fetch_row = [&](__amdgpu_buffer_rsrc_t r, unsigned off, f32x4 (&a4)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) a4[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 16 * k, 0));
  };

  for (int base = sr.off + (blockIdx.x * (PG_THREADS / 64) + wave) * 64; base < end; base += gridDim.x * (PG_THREADS / 64) * 64) {
    const int i = base + lane;
    const uint32_t key = i < end ? p.key_sorted[i] : 0xffffffffu;
    const bool head = i < end && (i == sr.off || p.key_sorted[i - 1] != key);
    const int mycell = (int)(key - (uint32_t)b * (uint32_t)ncell);
    const int my_y = mycell / p.W, my_x = mycell - my_y * p.W;
    const int mycls = (((my_y + 1) & 1) << 1) | ((my_x + 1) & 1);   // (ky0, kx0): first reachable tap per axis
    for (int cls = 0; cls < 4; ++cls) {
      unsigned long long m = __ballot(head && mycls == cls);
      const int ky0 = cls >> 1, kx0 = cls & 1;
      const int nky = ky0 ? 1 : 2, nkx = kx0 ? 1 : 2;
      while (m) {   // batches of up to 16 cells of this class
        const bool in = (m >> lane) & 1;
        const int rank = __popcll(m & ((1ull << lane) - 1));
        if (in && rank < 16) rowcell[rank] = mycell;
        const int nrows = min(16, (int)__popcll(m));
        m = __ballot(in && rank >= 16);
        lds_fence();
        const int cell = li < nrows ? rowcell[li] : -1;
        int crow[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) crow[r] = (4 * lq + r) < nrows ? rowcell[4 * lq + r] : -1;
        lds_fence();
        const int y = cell / p.W, x = cell - y * p.W;
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        f32x4 a_sk[4], a_t[4][4];
        fetch_row(dskr, cell >= 0 ? (unsigned)((cell * p.dskip.ld + 16 * lq) * 4) : OOB, a_sk);
#pragma unroll
        for (int t = 0; t < 4; ++t) {   // tap t = (iy, ix) = (t >> 1, t & 1); taps beyond (nky, nkx) are not fetched
          const int iy = t >> 1, ix = t & 1;
          if (iy < nky && ix < nkx) {
            const int ky = ky0 + 2 * iy, kx = kx0 + 2 * ix;
            const int oy = (y + 1 - ky) >> 1, ox = (x + 1 - kx) >> 1;
            const bool ok = cell >= 0 && oy >= 0 && oy < h2 && ox >= 0 && ox < w2;
            fetch_row(dyr, ok ? (unsigned)(((oy * w2 + ox) * 64 + 16 * lq) * 4) : OOB, a_t[t]);
          }
        }
        mma_row(a_sk, W3s, acc);
#pragma unroll
        for (int t = 0; t < 4; ++t) {   // the same tap order as before: (iy, ix) ascending
          const int iy = t >> 1, ix = t & 1;
          if (iy < nky && ix < nkx) mma_row(a_t[t], W1s + ((ky0 + 2 * iy) * 3 + kx0 + 2 * ix) * 64 * 32, acc);
        }
        // C layout: acc[nt][r] = row 4 lq + r, channel 16 nt + li
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (crow[r] < 0) continue;
          float* o = out + (int64_t)crow[r] * p.dcanvas.ld + li;
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) o[16 * nt] = (p.accumulate ? o[16 * nt] : 0.f) + acc[nt][r];
        }
      }
    }
  }
#endif
}


// ---- pillar_input_grad on the 16-bit matrix pipe, full batches (round 6, second session) ----------------------------------------
// The form above spent its time in three places (B = 16: 0.25 ms per cloud for 22 GFLOP): batches a THIRD empty (a 64-point window
// holds ~9 pillar heads per parity class: 16-row MFMA batches 56 % full), fp32 MFMAs (v_mfma_f32_16x16x4_f32: 32 cycles each, 43 % of
// the launch) fed by scalar LDS weight reads, and one load round trip + one read-modify-write round trip per batch.  Here
//  * every wave keeps a circular QUEUE of pillar heads per parity class in LDS and multiplies only FULL batches of 16 (the tails once,
//    at the end of its range);
//  * the products are bf16x3 -- both operands as three bf16 planes (8 + 8 + 8 mantissa bits: the fp32 value exactly), six
//    v_mfma_f32_16x16x32_bf16 per 32-deep k step, the terms below 2^-24 dropped: fp32-rounding-exact like conv.hip's NP = 3 form, at
//    a sixth of the fp32 pipe's cycles; the weights of both convs are split ONCE per workgroup into LDS planes
//    [tap 0..8 | skip][plane][k step][c (32)][64 B] (120 KB), a B fragment = one ds_read_b128 (slot swizzle g, see sparse_conv3x3_h2_kernel);
//  * the old output values (accumulate) are fetched WITH the batch's gradient rows.
constexpr int PGQ = 128;                                  // queue capacity per wave and class (<= 15 left over + 64 new)
constexpr int PG3_WT = 10 * 3 * 2 * 32 * 16;              // weight planes, floats
typedef __bf16 bf16x8p_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void pg_split3(const f32x4 lo4, const f32x4 hi4, bf16x8p_t& h, bf16x8p_t& m, bf16x8p_t& l) {
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float v = k < 4 ? lo4[k & 3] : hi4[k & 3];
    h[k] = (__bf16)v;
    const float r1 = v - (float)h[k];
    m[k] = (__bf16)r1;
    l[k] = (__bf16)(r1 - (float)m[k]);
  }
}

__global__ __launch_bounds__(PG_THREADS) void pillar_input_grad_x3_kernel(PillarGradParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Wt = lds;                                                   // [10][3][2][32][16 floats]
  int* Queue = reinterpret_cast<int*>(Wt + PG3_WT);                  // [16 waves][4 classes][PGQ]
  const int tid = threadIdx.x;
  const int b = blockIdx.y, g = p.cloud;
  // ---- weights -> three bf16 planes.  piece i = (tap, k step, c, slot): the eight k values co = 32 ks + 8 slot + e of output channel c
  for (int i = tid; i < 10 * 2 * 32 * 4; i += PG_THREADS) {
    const int slot = i & 3, c = (i >> 2) & 31, ks = (i >> 7) & 1, tap = i >> 8;
    f32x4 v0, v1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int co = 32 * ks + 8 * slot + e;
      const float w = tap < 9 ? p.w1[(co * 9 + tap) * 32 + c] : p.w3[co * 64 + 32 * g + c];
      if (e < 4) v0[e] = w; else v1[e - 4] = w;
    }
    bf16x8p_t h, m, l;
    pg_split3(v0, v1, h, m, l);
    const int ps = slot ^ sp_g4((c >> 2) & 3);
    float* d = Wt + (((tap * 3 + 0) * 2 + ks) * 32 + c) * 16 + ps * 4;
    *reinterpret_cast<bf16x8p_t*>(d) = h;
    *reinterpret_cast<bf16x8p_t*>(d + 2 * 32 * 16) = m;
    *reinterpret_cast<bf16x8p_t*>(d + 2 * 2 * 32 * 16) = l;
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6, li = lane & 15, lq = lane >> 4;
  int* queue = Queue + wave * 4 * PGQ;
  const int ncell = p.H * p.W, h2 = p.H >> 1, w2 = p.W >> 1;
  const float* dy1 = p.dy1 + (int64_t)(g * p.B + b) * h2 * w2 * 64;
  const float* dsk = reinterpret_cast<const float*>(p.dskip.ptr) + df_img_base(p.dskip, b);
  float* out = reinterpret_cast<float*>(p.dcanvas.ptr) + df_img_base(p.dcanvas, b);
  const SampleRange sr = sample_range(p.counts, b);
  const int end = sr.off + sr.cnt;
  auto lds_fence = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  constexpr unsigned OOB = 0xF0000000u;
  const __amdgpu_buffer_rsrc_t dskr = __builtin_amdgcn_make_buffer_rsrc((void*)dsk, 0, (unsigned)(((int64_t)(ncell - 1) * p.dskip.ld + 64) * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t dyr = __builtin_amdgcn_make_buffer_rsrc((void*)dy1, 0, (unsigned)((int64_t)h2 * w2 * 64 * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t outr = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, (unsigned)(((int64_t)(ncell - 1) * p.dcanvas.ld + 32) * 4), 0x00020000);
  // a row's operand: channels 32 ks + 8 lq .. + 7 (k step ks) -> a4[2 ks], a4[2 ks + 1]
  auto fetch_row = [&](__amdgpu_buffer_rsrc_t r, unsigned off, f32x4 (&a4)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) a4[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 128 * (k >> 1) + 16 * (k & 1), 0));
  };
  const float* wlane = Wt + (li * 4 + (lq ^ sp_g4((li >> 2) & 3))) * 4;   // row c = li (+ 16 nt) of a [32][64 B] tile, this lane's slot
  auto mma_row = [&](const f32x4 (&a4)[4], int tap, f32x4 (&acc)[2]) {
    const float* wt = wlane + tap * (3 * 2 * 32 * 16);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8p_t ah, am, al;
      pg_split3(a4[2 * ks], a4[2 * ks + 1], ah, am, al);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const float* wp = wt + (ks * 32 + 16 * nt) * 16;
        const bf16x8p_t bh = *reinterpret_cast<const bf16x8p_t*>(wp);
        const bf16x8p_t bm = *reinterpret_cast<const bf16x8p_t*>(wp + 2 * 32 * 16);
        const bf16x8p_t bl = *reinterpret_cast<const bf16x8p_t*>(wp + 2 * 2 * 32 * 16);
        f32x4 c = acc[nt];
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);   // small terms first
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
        acc[nt] = c;
      }
    }
  };
  // one batch: rows 0 .. nrows - 1 of class cls's queue, starting at its head qh
  auto process = [&](int cls, int qh, int nrows) {
    const int* qc = queue + cls * PGQ;
    const int ky0 = cls >> 1, kx0 = cls & 1;
    const int nky = ky0 ? 1 : 2, nkx = kx0 ? 1 : 2;
    const int cell = li < nrows ? qc[(qh + li) & (PGQ - 1)] : -1;
    int crow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) crow[r] = (4 * lq + r) < nrows ? qc[(qh + 4 * lq + r) & (PGQ - 1)] : -1;
    const int y = cell / p.W, x = cell - y * p.W;
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    // rows of the batch: the skip gradient, then the class's reachable taps (iy, ix) ascending -- ONE row ahead of the products (all
    // five rows in flight at once, as the fp32 form has them, is 80 registers beside the three-plane fragments: 133 spilled)
    f32x4 cur[4], nxt[4];
    fetch_row(dskr, cell >= 0 ? (unsigned)((cell * p.dskip.ld + 8 * lq) * 4) : OOB, cur);
    float old[4][2];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        old[r][nt] = (p.accumulate && crow[r] >= 0)
                         ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(outr, (unsigned)((crow[r] * p.dcanvas.ld + 16 * nt + li) * 4), 0, 0)) : 0.f;
    const int ntap = nky * nkx;
    // row j of the batch: j = 0 the skip gradient (weights "tap" 9), j = 1 .. ntap the class's taps (iy, ix) ascending; TWO rows ahead of
    // the products (a row's products are ~0.3 us, its load round trip 1-2 us)
    auto tap_of = [&](int k, int& tapid) -> unsigned {     // offset of tap k's row (k < ntap) or OOB
      const int iy = nkx == 2 ? k >> 1 : k, ix = nkx == 2 ? k & 1 : 0;
      const int ky = ky0 + 2 * iy, kx = kx0 + 2 * ix;
      const int oy = (y + 1 - ky) >> 1, ox = (x + 1 - kx) >> 1;
      const bool ok = cell >= 0 && oy >= 0 && oy < h2 && ox >= 0 && ox < w2;
      tapid = ky * 3 + kx;
      return ok ? (unsigned)(((oy * w2 + ox) * 64 + 8 * lq) * 4) : OOB;
    };
    f32x4 nx2[4];
    int tap_cur = 9, tap_nxt, tap_nx2 = 0;
    fetch_row(dyr, tap_of(0, tap_nxt), nxt);               // (a class has at least one tap)
#pragma unroll 1
    for (int j = 0; j <= ntap; ++j) {
      if (j + 2 <= ntap) fetch_row(dyr, tap_of(j + 1, tap_nx2), nx2);
      mma_row(cur, tap_cur, acc);
#pragma unroll
      for (int q = 0; q < 4; ++q) { cur[q] = nxt[q]; nxt[q] = nx2[q]; }
      tap_cur = tap_nxt;
      tap_nxt = tap_nx2;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (crow[r] < 0) continue;
      float* o = out + (int64_t)crow[r] * p.dcanvas.ld + li;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) o[16 * nt] = old[r][nt] + acc[nt][r];
    }
  };

  int qh = 0, qn = 0;       // four 8-bit fields each: head index (mod PGQ) and fill of the class queues
  for (int base = sr.off + (blockIdx.x * (PG_THREADS / 64) + wave) * 64; base < end; base += gridDim.x * (PG_THREADS / 64) * 64) {
    const int i = base + lane;
    const uint32_t key = i < end ? p.key_sorted[i] : 0xffffffffu;
    const bool head = i < end && (i == sr.off || p.key_sorted[i - 1] != key);
    const int mycell = (int)(key - (uint32_t)b * (uint32_t)ncell);
    const int my_y = mycell / p.W, my_x = mycell - my_y * p.W;
    const int mycls = (((my_y + 1) & 1) << 1) | ((my_x + 1) & 1);
#pragma unroll
    for (int cls = 0; cls < 4; ++cls) {
      const unsigned long long m = __ballot(head && mycls == cls);
      const int h_c = (qh >> (8 * cls)) & 255, n_c = (qn >> (8 * cls)) & 255;
      if (head && mycls == cls) queue[cls * PGQ + ((h_c + n_c + (int)__popcll(m & ((1ull << lane) - 1))) & (PGQ - 1))] = mycell;
      qn += (int)__popcll(m) << (8 * cls);
    }
    lds_fence();
#pragma unroll 1
    for (int cls = 0; cls < 4; ++cls) {
      while (((qn >> (8 * cls)) & 255) >= 16) {
        process(cls, (qh >> (8 * cls)) & 255, 16);
        const int h_c = ((qh >> (8 * cls)) + 16) & (PGQ - 1);
        qh = (qh & ~(255 << (8 * cls))) | (h_c << (8 * cls));
        qn -= 16 << (8 * cls);
      }
    }
    lds_fence();       // (the batches' queue reads are behind us before the next window appends)
  }
#pragma unroll 1
  for (int cls = 0; cls < 4; ++cls) {
    const int n_c = (qn >> (8 * cls)) & 255;
    if (n_c > 0) process(cls, (qh >> (8 * cls)) & 255, n_c);
  }
#endif
}


// ------------------------------------------------------------------------------ sparse-output-gradient weight grad ---
// The UNet's last conv (3x3, 64 -> 64 at full resolution) receives its output gradient from the decoder's gather
// backward: exact zeros everywhere except at the cells pc0 points looked up (~20 % of H*W).  Its weight gradient
//   dW[co, ky, kx, ci] = sum_p dy[p, co] x[p + (ky - 1, kx - 1), ci]
// therefore only needs the occupied pixels p: 5x fewer FLOPs than the dense kernel (2.5 ms per step).  One wave per
// tap: every wave of a workgroup scans the same 64-point windows of the sorted pillar keys, ballots the pillar heads into
// its own pixel list and accumulates its tap's [64 co x 64 ci] block as 16 x (16 x 16 x 4) MFMA tiles, four pixels per
// step, both operands read straight from global memory (64-byte row segments, shared through L1/L2 by the nine waves).
// No workgroup synchronisation at all.  Partials [workgroup][64][9][64] are summed by df_conv2d_wgrad_reduce (fixed order);
// the centre-tap wave also produces the bias gradient partial.
struct SparseWgradParams {
  const uint32_t* key_sorted;
  const int32_t* counts;
  int H, W;
  df_img dy, x;
  float* ws;        // [gridDim.y * gridDim.x][64][9][64]
  float* bias_ws;   // [gridDim.y * gridDim.x][64] or nullptr
};

// Round 4 (second session): the kernel was neither matrix- nor bandwidth-bound (62 GFLOP at 53 TFLOP/s, the fp32 matrix pipe 40 %
// busy): a wave issued the 32 loads of a 16-pixel step, waited for them, then issued its 64 MFMAs, and a 64-point window holds
// ~35 pillars -- three steps with the last one a third full.  Now a window is 256 points (four ballots: ~140 pillars, nine
// steps, the last one wasted on average half a step instead of a third of every third), and the loads of step s + 1 are
// issued BEFORE the MFMAs of step s (two register sets, the loop unrolled by two): load latency rides under ~2000 matrix cycles.
constexpr int SW_WIN = 256;   // sorted points per window

__global__ __launch_bounds__(576) void sparse_wgrad3x3_kernel(SparseWgradParams p) {
#if defined(__HIP_DEVICE_COMPILE__)  // buffer-resource builtins only exist in the device pass
  __shared__ int Plist[9 * SW_WIN];
  const int tid = threadIdx.x, lane = tid & 63, tap = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int b = blockIdx.y, ncell = p.H * p.W;
  const int ty = tap / 3 - 1, tx = tap % 3 - 1;
  int* plist = Plist + tap * SW_WIN;
  const float* dy = reinterpret_cast<const float*>(p.dy.ptr) + df_img_base(p.dy, b);
  const float* xp = reinterpret_cast<const float*>(p.x.ptr) + df_img_base(p.x, b);
  // one sample's image is far below 4 GB (the C entry checks it): byte offsets fit the buffer instructions' 32 bits
  const __amdgpu_buffer_rsrc_t dyr = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, (unsigned)((int64_t)ncell * p.dy.ld * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)xp, 0, (unsigned)((int64_t)ncell * p.x.ld * 4), 0x00020000);
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  const SampleRange sr = sample_range(p.counts, b);
  const int end = sr.off + sr.cnt;
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (int base = sr.off + blockIdx.x * SW_WIN; base < end; base += gridDim.x * SW_WIN) {
    const int wend = min(end, base + SW_WIN);
    uint32_t key[SW_WIN / 64], prev[SW_WIN / 64];
#pragma unroll
    for (int w = 0; w < SW_WIN / 64; ++w) {   // all key loads of the window in flight together
      const int i = base + 64 * w + lane;
      key[w] = i < wend ? p.key_sorted[i] : 0xffffffffu;
      prev[w] = (i < wend && i > sr.off) ? p.key_sorted[i - 1] : 0xfffffffeu;
    }
    int n = 0;
#pragma unroll
    for (int w = 0; w < SW_WIN / 64; ++w) {
      const int i = base + 64 * w + lane;
      const bool head = i < wend && (i == sr.off || prev[w] != key[w]);
      const unsigned long long m = __ballot(head);
      if (head) plist[n + (int)__popcll(m & lt)] = (int)(key[w] - (uint32_t)b * (uint32_t)ncell);
      n += (int)__popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // one step = 16 pixels = four MFMA k steps: 32 loads, 64 MFMAs
    // branch-free: buffer loads whose offset is out of range for a missing pixel / a tap outside the image return 0 (a
    // conditional load compiles to a branch per load and a wait where its value merges with the 0)
    auto load_step = [&](int s0, float (&a)[4][4], float (&bv)[4][4]) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = s0 + 4 * u + lq;
        const int cell = k < n ? plist[k] : -1;
        const int y = cell / p.W, x = cell - y * p.W;
        const int qy = y + ty, qx = x + tx;
        const bool okq = cell >= 0 && qy >= 0 && qy < p.H && qx >= 0 && qx < p.W;
        const unsigned ao = cell >= 0 ? (unsigned)((cell * p.dy.ld + li) * 4) : 0xF0000000u;
        const unsigned bo = okq ? (unsigned)(((qy * p.W + qx) * p.x.ld + li) * 4) : 0xF0000000u;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          a[u][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dyr, ao, 64 * t, 0));
          bv[u][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, bo, 64 * t, 0));
        }
      }
    };
    auto mma_step = [&](const float (&a)[4][4], const float (&bv)[4][4]) {
      if (tap == 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int t = 0; t < 4; ++t) bsum[t] += a[u][t];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
            acc[ct][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][ct], bv[u][nt], acc[ct][nt], 0, 0, 0);
    };
    float a0[4][4], b0[4][4], a1[4][4], b1[4][4];
    if (n > 0) load_step(0, a0, b0);
    for (int s0 = 0; s0 < n; s0 += 32) {
      const bool more1 = s0 + 16 < n;
      if (more1) load_step(s0 + 16, a1, b1);
      mma_step(a0, b0);
      if (more1) {
        if (s0 + 32 < n) load_step(s0 + 32, a0, b0);
        mma_step(a1, b1);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  const int64_t blk = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
  float* o = p.ws + blk * 64 * 9 * 64;
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[((16 * ct + 4 * lq + r) * 9 + tap) * 64 + 16 * nt + li] = acc[ct][nt][r];
  if (tap == 4 && p.bias_ws) {   // bsum[t] on lane (li, lq) = sum over this lane's pixels of dy[.., 16 t + li]
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float v = bsum[t];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (lq == 0) p.bias_ws[blk * 64 + 16 * t + li] = v;
    }
  }
#endif
}


// ---- bf16x2 form of the sparse-output-gradient weight gradient (round 6, second session) ---------------------------------------
// sparse_wgrad3x3_kernel above multiplies on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32, four pixels per instruction): 62 GFLOP per
// step at 68 TFLOP/s.  Here both operands are two bf16 planes (hi = bf16(v), lo = bf16(v - hi): 16 significant bits, no scales -- the
// product of the GRU kernels, gemm_dma.h) and a k step is 32 PIXELS of one v_mfma_f32_16x16x32_bf16: three products per tile (lo hi',
// hi lo', hi hi'; small terms first) into ONE accumulator -- 48 instructions of 16 cycles per 32 pixels and tap instead of 128 of 32.
// Tile (ct, nt) row m <-> output channel 4 m + ct, column n <-> input channel 4 n + nt: a lane's operand values for the four row
// (column) tiles are then 16 contiguous bytes of one pixel's dy (x) row -- 16 dwordx4 loads per lane and step instead of 64 dword
// loads -- and its four column tiles of one output row are 16 contiguous bytes of the partial.  Same pixel lists, one wave per tap, no
// workgroup synchronisation, partials [workgroup][64][9][64] for df_conv2d_wgrad_reduce as before.
typedef __bf16 bf16x8s_t __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(576) void sparse_wgrad3x3_x2_kernel(SparseWgradParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ int Plist[9 * SW_WIN];
  const int tid = threadIdx.x, lane = tid & 63, tap = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int b = blockIdx.y, ncell = p.H * p.W;
  const int ty = tap / 3 - 1, tx = tap % 3 - 1;
  int* plist = Plist + tap * SW_WIN;
  const float* dy = reinterpret_cast<const float*>(p.dy.ptr) + df_img_base(p.dy, b);
  const float* xp = reinterpret_cast<const float*>(p.x.ptr) + df_img_base(p.x, b);
  const __amdgpu_buffer_rsrc_t dyr = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, (unsigned)((int64_t)ncell * p.dy.ld * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)xp, 0, (unsigned)((int64_t)ncell * p.x.ld * 4), 0x00020000);
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  const SampleRange sr = sample_range(p.counts, b);
  const int end = sr.off + sr.cnt;
  const unsigned long long lt = (1ull << lane) - 1ull;
  typedef unsigned u32x4s_t __attribute__((ext_vector_type(4)));
  typedef float f32x2s_t __attribute__((ext_vector_type(2)));
  auto split8 = [&](const f32x4 (&r)[8], int c, bf16x8s_t& hi, bf16x8s_t& lo) {   // component c of the eight pixels' values
#pragma unroll
    for (int e = 0; e < 8; e += 2) {      // (pairs: the residual as ONE packed subtraction -- this file is built without the SLP vectoriser)
      const f32x2s_t v = {r[e][c], r[e + 1][c]};
      hi[e] = (__bf16)v[0];
      hi[e + 1] = (__bf16)v[1];
      const f32x2s_t h = {(float)hi[e], (float)hi[e + 1]};
      const f32x2s_t d = v - h;
      lo[e] = (__bf16)d[0];
      lo[e + 1] = (__bf16)d[1];
    }
  };
  for (int base = sr.off + blockIdx.x * SW_WIN; base < end; base += gridDim.x * SW_WIN) {
    const int wend = min(end, base + SW_WIN);
    uint32_t key[SW_WIN / 64], prev[SW_WIN / 64];
#pragma unroll
    for (int w = 0; w < SW_WIN / 64; ++w) {
      const int i = base + 64 * w + lane;
      key[w] = i < wend ? p.key_sorted[i] : 0xffffffffu;
      prev[w] = (i < wend && i > sr.off) ? p.key_sorted[i - 1] : 0xfffffffeu;
    }
    int n = 0;
#pragma unroll
    for (int w = 0; w < SW_WIN / 64; ++w) {
      const int i = base + 64 * w + lane;
      const bool head = i < wend && (i == sr.off || prev[w] != key[w]);
      const unsigned long long m = __ballot(head);
      if (head) {      // (y << 16 | x: ONE division per pillar here instead of one per pixel, step and tap below -- 25 VALU instructions each)
        const int cell = (int)(key[w] - (uint32_t)b * (uint32_t)ncell);
        const int y = cell / p.W;
        plist[n + (int)__popcll(m & lt)] = (y << 16) | (cell - y * p.W);
      }
      n += (int)__popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 1
    for (int s0 = 0; s0 < n; s0 += 32) {
      // this lane's eight pixels of the step: k = 8 lq + e.  Branch-free: a missing pixel / a tap outside the image reads zeros from an
      // out-of-range offset
      f32x4 ra[8], rb[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = s0 + 8 * lq + e;
        const int yx = plist[min(k, n - 1)];           // (n >= 1 here; branch-free: the entry of a missing pixel is not used)
        const bool have = k < n;
        const int y = yx >> 16, x = yx & 0xffff;
        const int cell = y * p.W + x;
        const int qy = y + ty, qx = x + tx;
        const bool okq = have && qy >= 0 && qy < p.H && qx >= 0 && qx < p.W;
        const unsigned ao = have ? (unsigned)((cell * p.dy.ld + 4 * li) * 4) : 0xF0000000u;
        const unsigned bo = okq ? (unsigned)(((qy * p.W + qx) * p.x.ld + 4 * li) * 4) : 0xF0000000u;
        ra[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dyr, ao, 0, 0));
        rb[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, bo, 0, 0));
      }
      if (tap == 4) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
          for (int c = 0; c < 4; ++c) bsum[c] += ra[e][c];
      }
      bf16x8s_t ah[4], al[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) split8(ra, c, ah[c], al[c]);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        bf16x8s_t bh, bl;
        split8(rb, nt, bh, bl);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          f32x4 v = acc[ct][nt];
          v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[ct], bh, v, 0, 0, 0);   // small terms first
          v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[ct], bl, v, 0, 0, 0);
          v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[ct], bh, v, 0, 0, 0);
          acc[ct][nt] = v;
        }
        __builtin_amdgcn_sched_barrier(0);   // (one column tile's fragments at a time: hoisting all four splits spilled 23 registers)
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  const int64_t blk = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
  float* o = p.ws + blk * 64 * 9 * 64;
  // acc[ct][nt][r] = dW[co = 4 (4 lq + r) + ct][tap][ci = 4 li + nt]
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      st4(o + ((4 * (4 * lq + r) + ct) * 9 + tap) * 64 + 4 * li, f32x4{acc[ct][0][r], acc[ct][1][r], acc[ct][2][r], acc[ct][3][r]});
  if (tap == 4 && p.bias_ws) {   // bsum[c] on lane (li, lq) = sum over this lane's pixels of dy[.., 4 li + c]
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float v = bsum[c];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      bsum[c] = v;
    }
    if (lq == 0) st4(p.bias_ws + blk * 64 + 4 * li, f32x4{bsum[0], bsum[1], bsum[2], bsum[3]});
  }
#endif
}


// ----------------------------------------------------------------------------------- sparse-output 3x3 convolution ---
// The UNet's last conv (3x3, stride 1, 64 -> 64, full resolution) produces the `after` image, which is only ever READ
// at the cells pc0 points look up (the decoder's gather; ~20 % of H*W).  This kernel computes
//   y[p, co] = bias[co] + sum_{ky,kx} sum_ci x[p + (ky - 1, kx - 1), ci] w[co, ky, kx, ci]
// for those cells only: a wave scans 64 consecutive sorted points, ballots the pillar heads and multiplies batches of 16
// cells [16 x 576] by the transposed weights [576 x 64] (LDS, 147 KB: one 16-wave workgroup per CU) with
// v_mfma_f32_16x16x4_f32; A rows come straight from global memory (see pillar_input_grad_kernel).  Other cells of y are
// not written.  2.5 ms -> 0.4 ms per training step, and 3 % of a B=1 inference.
struct SparseConvParams {
  const uint32_t* key_sorted;
  const int32_t* counts;
  int H, W;
  df_img x, y;
  const float* w;      // [64][3][3][64]
  const float* bias;   // [64] or nullptr
};

__global__ __launch_bounds__(PG_THREADS) void sparse_conv3x3_kernel(SparseConvParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // B operand layout: Wt[tap][q][co][16 s] -- the 16 k values (ci = 16 q + s) one lane needs for an output channel are
  // contiguous, so a tap costs 16 ds_read_b128 per lane instead of 64 ds_read_b32; the four 16-byte slots of a row are
  // XOR-swizzled by (co & 3) to halve the bank conflicts of the 64-byte row pitch
  float* Wt = lds;                                              // [9][4][64][16]
  int* RowCell = reinterpret_cast<int*>(Wt + 9 * 64 * 64);     // [16 waves][16]
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  {  // weights [co][tap][ci]: nine 16-byte loads per thread issued together; (ci..ci+3) = one slot of row (tap, q, co)
    f32x4 wv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wv[k] = ld4(p.w + (tid + PG_THREADS * k) * 4);
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int i = (tid + PG_THREADS * k) * 4;
      const int ci = i & 63, tap = (i >> 6) % 9, co = i / 576;
      const int q = ci >> 4, slot = (ci >> 2) & 3;
      st4(Wt + (((tap * 4 + q) * 64 + co) * 4 + (slot ^ (co & 3))) * 4, wv[k]);
    }
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6, li = lane & 15, lq = lane >> 4;
  int* rowcell = RowCell + wave * 16;
  const int ncell = p.H * p.W;
  const float* xp = reinterpret_cast<const float*>(p.x.ptr) + df_img_base(p.x, b);
  float* yp = reinterpret_cast<float*>(p.y.ptr) + df_img_base(p.y, b);
  const SampleRange sr = sample_range(p.counts, b);
  const int end = sr.off + sr.cnt;
  float bia[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) bia[nt] = p.bias ? p.bias[16 * nt + li] : 0.f;
  auto lds_fence = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  // contiguous chunk range per workgroup (not grid-strided; its waves take the chunks of the range in turn): the sorted
  // order is row-major, so a workgroup walks a band of image rows and the x rows of its vertical taps stay in its XCD's
  // L2 (758 -> 680 us; the same change made sparse_wgrad3x3_kernel 4 % slower and was not kept there)
  const int nchunk = (sr.cnt + 63) / 64, per = (nchunk + (int)gridDim.x - 1) / (int)gridDim.x;
  const int c_lo = blockIdx.x * per, c_hi = min(c_lo + per, nchunk);
  for (int base = sr.off + (c_lo + wave) * 64; base < sr.off + c_hi * 64; base += (PG_THREADS / 64) * 64) {
    const int i = base + lane;
    const uint32_t key = i < end ? p.key_sorted[i] : 0xffffffffu;
    const bool head = i < end && (i == sr.off || p.key_sorted[i - 1] != key);
    const int mycell = (int)(key - (uint32_t)b * (uint32_t)ncell);
    unsigned long long m = __ballot(head);
    while (m) {   // batches of up to 16 cells
      const bool in = (m >> lane) & 1;
      const int rank = __popcll(m & ((1ull << lane) - 1));
      if (in && rank < 16) rowcell[rank] = mycell;
      const int nrows = min(16, (int)__popcll(m));
      m = __ballot(in && rank >= 16);
      lds_fence();
      const int cell = li < nrows ? rowcell[li] : -1;
      int crow[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) crow[r] = (4 * lq + r) < nrows ? rowcell[4 * lq + r] : -1;
      lds_fence();
      const int y = cell / p.W, x = cell - y * p.W;
      f32x4 acc[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{bia[nt], bia[nt], bia[nt], bia[nt]};
      // A rows of tap t + 1 are fetched while tap t is multiplied
      auto fetch = [&](int tap, f32x4 (&a4)[4]) {
        const int qy = y + tap / 3 - 1, qx = x + tap % 3 - 1;
        const bool ok = cell >= 0 && qy >= 0 && qy < p.H && qx >= 0 && qx < p.W;
        const float* src = xp + (int64_t)(ok ? qy * p.W + qx : 0) * p.x.ld + 16 * lq;
#pragma unroll
        for (int k = 0; k < 4; ++k) a4[k] = ok ? ld4(src + 4 * k) : f32x4{0.f, 0.f, 0.f, 0.f};
      };
      f32x4 a_cur[4], a_nxt[4];
      fetch(0, a_cur);
#pragma unroll 1
      for (int tap = 0; tap < 9; ++tap) {
        if (tap + 1 < 9) fetch(tap + 1, a_nxt);
        const float* wrow = Wt + ((tap * 4 + lq) * 64 + li) * 16;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const float* wr = wrow + nt * 16 * 16;            // co = 16 nt + li; (co & 3) == (li & 3)
          f32x4 w4[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) w4[k] = ld4(wr + ((k ^ (li & 3)) * 4));
#pragma unroll
          for (int s2 = 0; s2 < 16; ++s2)
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[s2 >> 2][s2 & 3], w4[s2 >> 2][s2 & 3], acc[nt], 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) a_cur[k] = a_nxt[k];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (crow[r] < 0) continue;
        float* o = yp + (int64_t)crow[r] * p.y.ld + li;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) o[16 * nt] = acc[nt][r];
      }
    }
  }
}


// ---- fp16x2 form of the sparse-output 3x3 convolution (round 6, second session) ----------------------------------------------
// The fp32 form above runs v_mfma_f32_16x16x4_f32: 62 GFLOP per step at 81 TFLOP/s, half of the fp32 matrix pipe's peak -- the last
// dense-rate fp32 MFMA kernel in the forward.  Here the product is the fp16x2 one of the dense 3x3 layers (conv.hip: two scaled fp16
// planes per operand, x s = hi + lo / 2048, THREE v_mfma_f32_16x16x32_f16 per 32-deep k step -- hi hi' in one accumulator, the two cross
// terms in a second one that is folded in at the end; 22 significant bits per operand): 24 products of 16 cycles per tap and batch
// instead of 64 of 32.  The weights arrive pre-split (df_split_h2 / df_weight_prep planes [hi | lo], each [co][tap][ci] fp16) and sit in
// LDS as [tap][plane][k step][co][64 B] -- the same 147 KB as the fp32 form -- a B fragment is one ds_read_b128 (16-byte slots swizzled
// by g((co >> 2) & 3), g = (0, 2, 3, 1): conflict-free for the lane groups ds_read_b128 is serviced in, see gemm_dma.h); the A rows come
// straight from global memory as before (lane (cell, q) holds channels 8 q .. 8 q + 7 and 32 + 8 q .. of its cell's row) and are split in
// registers with the scale of the input's bound *amax_x (the producing convolution's measured max |u|).
struct SparseConvH2Params {
  const uint32_t* key_sorted;
  const int32_t* counts;
  int H, W;
  df_img x, y;
  const void* w2;      // [2][64][3][3][64] fp16: hi plane, lo plane of w s_w
  const float* bias;   // [64] or nullptr
  const float* amax_x;
  const float* amax_w;
};
typedef _Float16 f16x8s_t __attribute__((ext_vector_type(8)));

// B16 (bf16-operand training mode, BASELINE configs[4] "bf16 MFMA"): ONE bf16 plane per operand, one v_mfma_f32_16x16x32_bf16 per k step
// -- what the mode's other convolutions compute; p.w2 then points at the fp32 weights [co][tap][ci], rounded to bf16 as they enter LDS
// ([tap][k step][co][64 B], 73 KB), amax_x / amax_w unused.
template <bool B16>
__global__ __launch_bounds__(PG_THREADS) void sparse_conv3x3_h2_kernel(SparseConvH2Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Wt = lds;                                              // [9][2][2][64][16 floats = 64 B]
  int* RowCell = reinterpret_cast<int*>(Wt + 9 * 2 * 2 * 64 * 16);     // [16 waves][PGQ]: each wave's circular queue of pillar heads
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  if constexpr (B16) {   // 4608 pieces of eight consecutive ci: (co, tap, k step, slot) = the digits of i
    typedef __bf16 bf16x8w_t __attribute__((ext_vector_type(8)));
    const float* wf = reinterpret_cast<const float*>(p.w2);
    for (int i = tid; i < 4608; i += PG_THREADS) {
      const f32x4 v0 = ld4(wf + i * 8), v1 = ld4(wf + i * 8 + 4);
      bf16x8w_t h;
#pragma unroll
      for (int e = 0; e < 4; ++e) { h[e] = (__bf16)v0[e]; h[4 + e] = (__bf16)v1[e]; }
      const int co = i / 72, r2 = i - co * 72;
      const int tap = r2 >> 3, ks = (r2 >> 2) & 1, sl = r2 & 3;
      *reinterpret_cast<bf16x8w_t*>(Wt + (((tap * 2 + ks) * 64 + co) * 4 + (sl ^ sp_g4((co >> 2) & 3))) * 4) = h;
    }
  } else {  // 9216 16-byte pieces: piece i of the planes (linear in memory) = plane i / 4608, then (co, tap, k step, slot) = the digits of i % 4608
    f32x4 wv[9];
    const float* w2f = reinterpret_cast<const float*>(p.w2);
#pragma unroll
    for (int k = 0; k < 9; ++k) wv[k] = ld4(w2f + (tid + PG_THREADS * k) * 4);
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int i = tid + PG_THREADS * k;
      const int plane = i / 4608, rem = i - plane * 4608;
      const int co = rem / 72, r2 = rem - co * 72;
      const int tap = r2 >> 3, ks = (r2 >> 2) & 1, sl = r2 & 3;
      st4(Wt + ((((tap * 2 + plane) * 2 + ks) * 64 + co) * 4 + (sl ^ sp_g4((co >> 2) & 3))) * 4, wv[k]);
    }
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6, li = lane & 15, lq = lane >> 4;
  int* rowcell = RowCell + wave * PGQ;
  const int ncell = p.H * p.W;
  const float* xp = reinterpret_cast<const float*>(p.x.ptr) + df_img_base(p.x, b);
  float* yp = reinterpret_cast<float*>(p.y.ptr) + df_img_base(p.y, b);
  const SampleRange sr = sample_range(p.counts, b);
  const int end = sr.off + sr.cnt;
  float sx = 1.f, sw = 1.f;
  if constexpr (!B16) { sx = sp_h2_scale(*p.amax_x); sw = sp_h2_scale(*p.amax_w); }
  const float inv = (1.f / sx) * (1.f / sw);
  float bia[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) bia[nt] = p.bias ? p.bias[16 * nt + li] : 0.f;
  auto lds_fence = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  const float* wlane = Wt + (li * 4 + (lq ^ sp_g4((li >> 2) & 3))) * 4;     // this lane's slot of row co = li (+ 16 nt) of a [64][64 B] tile
  const int nchunk = (sr.cnt + 63) / 64, per = (nchunk + (int)gridDim.x - 1) / (int)gridDim.x;
  const int c_lo = blockIdx.x * per, c_hi = min(c_lo + per, nchunk);
  // FULL batches (round 6, second session): pillar heads go through a circular queue per wave and are multiplied sixteen at a time -- a
  // 64-point window holds ~35 heads, i.e. batches of 16, 16 and 3: a third of the products multiplied empty rows; the tail once, at the end
  auto process = [&](int qh, int nrows) {
    const int cell = li < nrows ? rowcell[(qh + li) & (PGQ - 1)] : -1;
    int crow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) crow[r] = (4 * lq + r) < nrows ? rowcell[(qh + 4 * lq + r) & (PGQ - 1)] : -1;
    const int y = cell / p.W, x = cell - y * p.W;
    f32x4 acc[4], acc1[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc1[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // A rows of tap t + 1 are fetched while tap t is multiplied: a4[2 ks], a4[2 ks + 1] = channels 32 ks + 8 lq .. + 7
    auto fetch = [&](int tap, f32x4 (&a4)[4]) {
      const int qy = y + tap / 3 - 1, qx = x + tap % 3 - 1;
      const bool ok = cell >= 0 && qy >= 0 && qy < p.H && qx >= 0 && qx < p.W;
      const float* src = xp + (int64_t)(ok ? qy * p.W + qx : 0) * p.x.ld + 8 * lq;
#pragma unroll
      for (int k = 0; k < 4; ++k) a4[k] = ok ? ld4(src + 32 * (k >> 1) + 4 * (k & 1)) : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    f32x4 a_cur[4], a_nxt[4];
    fetch(0, a_cur);
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
      if (tap + 1 < 9) fetch(tap + 1, a_nxt);
      if constexpr (B16) {
        typedef __bf16 bf16x8w_t __attribute__((ext_vector_type(8)));
        bf16x8w_t a8[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int e = 0; e < 8; ++e) a8[ks][e] = (__bf16)a_cur[2 * ks + (e >> 2)][e & 3];
        const float* wt = wlane + tap * (2 * 64 * 16);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8[ks], *reinterpret_cast<const bf16x8w_t*>(wt + (ks * 64 + 16 * nt) * 16), acc[nt], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) a_cur[k] = a_nxt[k];
        continue;
      }
      f16x8s_t ah[2], al[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float t = a_cur[2 * ks + (e >> 2)][e & 3] * sx;
          ah[ks][e] = (_Float16)t;
          al[ks][e] = (_Float16)((t - (float)ah[ks][e]) * 2048.f);
        }
      const float* wt = wlane + tap * (2 * 2 * 64 * 16);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const f16x8s_t bh = *reinterpret_cast<const f16x8s_t*>(wt + (ks * 64 + 16 * nt) * 16);
          const f16x8s_t bl = *reinterpret_cast<const f16x8s_t*>(wt + ((2 + ks) * 64 + 16 * nt) * 16);
          acc1[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[ks], bh, acc1[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ks], bh, acc[nt], 0, 0, 0);
          acc1[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ks], bl, acc1[nt], 0, 0, 0);
        }
#pragma unroll
      for (int k = 0; k < 4; ++k) a_cur[k] = a_nxt[k];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (crow[r] < 0) continue;
      float* o = yp + (int64_t)crow[r] * p.y.ld + li;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) o[16 * nt] = fmaf(fmaf(acc1[nt][r], 1.f / 2048.f, acc[nt][r]), inv, bia[nt]);
    }
  };
  int qh = 0, qn = 0;
  for (int base = sr.off + (c_lo + wave) * 64; base < sr.off + c_hi * 64; base += (PG_THREADS / 64) * 64) {
    const int i = base + lane;
    const uint32_t key = i < end ? p.key_sorted[i] : 0xffffffffu;
    const bool head = i < end && (i == sr.off || p.key_sorted[i - 1] != key);
    const unsigned long long m = __ballot(head);
    if (head) rowcell[(qh + qn + (int)__popcll(m & ((1ull << lane) - 1))) & (PGQ - 1)] = (int)(key - (uint32_t)b * (uint32_t)ncell);
    qn += (int)__popcll(m);
    lds_fence();
    while (qn >= 16) {
      process(qh, 16);
      qh = (qh + 16) & (PGQ - 1);
      qn -= 16;
    }
    lds_fence();
  }
  if (qn > 0) process(qh, qn);
#endif
}


// ---------------------------------------------------------------------------------- sparse-input weight gradient ---
// Weight gradient of the FIRST encoder conv (3x3, stride 2, pad 1, 32 -> 64) whose input is the pillar canvas: only the
// occupied cells q of a cloud contribute,
//   dW[co, ky, kx, ci] = sum_q [ (q + 1 - k) even and in range ] dy1[g*B+b, (q + 1 - k) / 2, co] canvas[b, q, 32 g + ci].
// One wave per tap (as sparse_wgrad3x3_kernel): every wave scans the same 64-point windows of the sorted pillar keys and
// keeps the heads whose parity admits its tap; four cells per MFMA k step, [64 co x 32 ci] accumulators per wave.
// Partials [workgroup][64][9][32]; the two clouds share the weights (second launch reduced with accumulate).
struct SparseInWgradParams {
  const uint32_t* key_sorted;
  const int32_t* counts;
  int B, H, W, cloud;
  const float* dy1;   // [2B][H/2][W/2][64]
  df_img canvas;      // [B][H][W][32] channel slice of the network input
  float* ws;          // [gridDim.y * gridDim.x][64][9][32]
};

// (Round 4, second session: windows of 512 sorted points -- a tap keeps about a quarter of a window's pillar heads, ~9 of a
// 64-point window: one or two 8-cell steps, each a load round trip in front of 16 MFMAs -- 16-cell steps whose loads are issued a
// step ahead of the MFMAs, branch-free buffer loads; as sparse_wgrad3x3_kernel above.)
constexpr int SIW_WIN = 512;

__global__ __launch_bounds__(576) void sparse_in_wgrad_kernel(SparseInWgradParams p) {
#if defined(__HIP_DEVICE_COMPILE__)  // buffer-resource builtins only exist in the device pass
  __shared__ int Plist[9 * SIW_WIN];
  const int tid = threadIdx.x, lane = tid & 63, tap = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int b = blockIdx.y, ncell = p.H * p.W, h2 = p.H >> 1, w2 = p.W >> 1;
  const int ky = tap / 3, kx = tap % 3;
  int* plist = Plist + tap * SIW_WIN;
  const float* dy1 = p.dy1 + (int64_t)(p.cloud * p.B + b) * h2 * w2 * 64;
  const float* cv = reinterpret_cast<const float*>(p.canvas.ptr) + df_img_base(p.canvas, b);
  const __amdgpu_buffer_rsrc_t dyr = __builtin_amdgcn_make_buffer_rsrc((void*)dy1, 0, (unsigned)((int64_t)h2 * w2 * 64 * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t cvr = __builtin_amdgcn_make_buffer_rsrc((void*)cv, 0, (unsigned)(((int64_t)(ncell - 1) * p.canvas.ld + 32) * 4), 0x00020000);
  f32x4 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const SampleRange sr = sample_range(p.counts, b);
  const int end = sr.off + sr.cnt;
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (int base = sr.off + blockIdx.x * SIW_WIN; base < end; base += gridDim.x * SIW_WIN) {
    const int wend = min(end, base + SIW_WIN);
    uint32_t key[SIW_WIN / 64], prev[SIW_WIN / 64];
#pragma unroll
    for (int w = 0; w < SIW_WIN / 64; ++w) {
      const int i = base + 64 * w + lane;
      key[w] = i < wend ? p.key_sorted[i] : 0xffffffffu;
      prev[w] = (i < wend && i > sr.off) ? p.key_sorted[i - 1] : 0xfffffffeu;
    }
    int n = 0;
#pragma unroll
    for (int w = 0; w < SIW_WIN / 64; ++w) {
      const int i = base + 64 * w + lane;
      const int cell = (int)(key[w] - (uint32_t)b * (uint32_t)ncell);
      const int y = cell / p.W, x = cell - y * p.W;
      const int oy = (y + 1 - ky) >> 1, ox = (x + 1 - kx) >> 1;
      const bool mine = i < wend && (i == sr.off || prev[w] != key[w]) && (((y + 1 - ky) & 1) == 0) &&
                        (((x + 1 - kx) & 1) == 0) && oy >= 0 && oy < h2 && ox >= 0 && ox < w2;
      const unsigned long long m = __ballot(mine);
      if (mine) plist[n + (int)__popcll(m & lt)] = (y << 16) | x;      // (packed: no division per cell and step below)
      n += (int)__popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // one step = 16 cells = four MFMA k steps: 12 loads, 32 MFMAs.  Round 6 (second session): tile (ct, nt) row m <-> output channel
    // 4 m + ct, column n <-> input channel 2 n + nt, so a lane's four row-tile values are ONE 16-byte load of its cell's dy1 row and
    // its two column-tile values two adjacent dwords of the canvas row (24 dword loads at 64-byte strides before); same products in the same order
    auto load_step = [&](int s0, float (&a)[4][4], float (&bv)[4][2]) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = s0 + 4 * u + lq;
        const int yx = plist[min(k, n - 1)];
        const bool have = k < n;
        const int y2 = yx >> 16, x2 = yx & 0xffff;
        const unsigned ao = have ? (unsigned)(((((y2 + 1 - ky) >> 1) * w2 + ((x2 + 1 - kx) >> 1)) * 64 + 4 * li) * 4) : 0xF0000000u;
        const unsigned bo = have ? (unsigned)(((y2 * p.W + x2) * p.canvas.ld + 2 * li) * 4) : 0xF0000000u;
        const f32x4 av = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dyr, ao, 0, 0));
#pragma unroll
        for (int t = 0; t < 4; ++t) a[u][t] = av[t];
        // (two dword loads of the pair.  The first cut loaded 8 bytes into a 2-vector w and took __builtin_bit_cast(float, w[0]) / (.., w[1]):
        //  __builtin_bit_cast of a vector ELEMENT reads element 0 whatever the index -- tools/repro/bitcast_vector_element.hip)
        bv[u][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(cvr, bo, 0, 0));
        bv[u][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(cvr, bo, 4, 0));
      }
    };
    auto mma_step = [&](const float (&a)[4][4], const float (&bv)[4][2]) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
            acc[ct][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][ct], bv[u][nt], acc[ct][nt], 0, 0, 0);
    };
    float a0[4][4], b0[4][2], a1[4][4], b1[4][2];
    if (n > 0) load_step(0, a0, b0);
    for (int s0 = 0; s0 < n; s0 += 32) {
      const bool more1 = s0 + 16 < n;
      if (more1) load_step(s0 + 16, a1, b1);
      mma_step(a0, b0);
      if (more1) {
        if (s0 + 32 < n) load_step(s0 + 32, a0, b0);
        mma_step(a1, b1);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  float* o = p.ws + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 64 * 9 * 32;
  // acc[ct][nt][r] = dW[co = 4 (4 lq + r) + ct][tap][ci = 2 li + nt]
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float* q = o + ((4 * (4 * lq + r) + ct) * 9 + tap) * 32 + 2 * li;
      q[0] = acc[ct][0][r];
      q[1] = acc[ct][1][r];
    }
#endif
}

// ---------------------------------------------------------------------------- cell sort -------------
// Counting sort of arbitrary cell keys (the stand-alone decoder-head call, decoder.pack_infos: voxel coordinates handed in by
// the caller, not produced by the pillariser): idx_sorted = the indices i of the keys < ncells grouped by key, ASCENDING i
// inside a group (= a stable sort, so the gather-backward's per-cell sums have one fixed order), cell_rng[k] = [start, end).
//   hist     cell_rng[k].end   = number of keys equal to k          (integer atomics: order-independent)
//   offsets  cell_rng[k].start = cell_rng[k].end = exclusive prefix  (2048 cells per workgroup + one pass over the block sums)
//   scatter  idx[cell_rng[k].end++] = i                              (arrival order inside a group ...)
//   order    insertion sort of each group with more than one member  (... made ascending: the result is deterministic)
constexpr int CS_CHUNK = 2048;

__global__ void cell_hist_kernel(const uint32_t* __restrict__ key, int64_t n, uint32_t ncells, int32_t* __restrict__ cell_rng) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t k = key[i];
  if (k < ncells) atomicAdd(&cell_rng[2 * (int64_t)k + 1], 1);
}

// exclusive scan of one int per thread across a 256-thread workgroup; returns the prefix, *total = the workgroup's sum
__device__ __forceinline__ int block_excl_scan256(int v, int* lds /*[256]*/, int* total) {
  const int t = threadIdx.x;
  lds[t] = v;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    const int a = t >= d ? lds[t - d] : 0;
    __syncthreads();
    lds[t] += a;
    __syncthreads();
  }
  const int incl = lds[t];
  *total = lds[255];
  __syncthreads();
  return incl - v;
}

__global__ __launch_bounds__(256) void cell_blocksum_kernel(const int32_t* __restrict__ cell_rng, int64_t ncells,
                                                            int32_t* __restrict__ blk_sum) {
  __shared__ int lds[256];
  const int64_t c0 = (int64_t)blockIdx.x * CS_CHUNK + threadIdx.x * 8;
  int v = 0;
  for (int j = 0; j < 8; ++j)
    if (c0 + j < ncells) v += cell_rng[2 * (c0 + j) + 1];
  int total;
  block_excl_scan256(v, lds, &total);
  if (threadIdx.x == 0) blk_sum[blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void cell_blockscan_kernel(int32_t* __restrict__ blk_sum, int nb) {
  __shared__ int lds[256];
  int carry = 0;
  for (int base = 0; base < nb; base += 256) {      // one workgroup walks the block sums (nb <= ncells / 2048)
    const int i = base + threadIdx.x;
    const int v = i < nb ? blk_sum[i] : 0;
    int total;
    const int ex = block_excl_scan256(v, lds, &total);
    if (i < nb) blk_sum[i] = carry + ex;
    carry += total;
  }
}

__global__ __launch_bounds__(256) void cell_offsets_kernel(int32_t* __restrict__ cell_rng, int64_t ncells,
                                                           const int32_t* __restrict__ blk_off) {
  __shared__ int lds[256];
  const int64_t c0 = (int64_t)blockIdx.x * CS_CHUNK + threadIdx.x * 8;
  int cnt[8], v = 0;
  for (int j = 0; j < 8; ++j) {
    cnt[j] = c0 + j < ncells ? cell_rng[2 * (c0 + j) + 1] : 0;
    v += cnt[j];
  }
  int total;
  int off = blk_off[blockIdx.x] + block_excl_scan256(v, lds, &total);
  for (int j = 0; j < 8; ++j) {
    if (c0 + j < ncells) {
      cell_rng[2 * (c0 + j)] = off;
      cell_rng[2 * (c0 + j) + 1] = off;       // the scatter's fill cursor; back at `end` when it is done
    }
    off += cnt[j];
  }
}

__global__ void cell_scatter_kernel(const uint32_t* __restrict__ key, int64_t n, uint32_t ncells, int32_t* __restrict__ cell_rng,
                                    uint32_t* __restrict__ idx_sorted) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t k = key[i];
  if (k < ncells) idx_sorted[atomicAdd(&cell_rng[2 * (int64_t)k + 1], 1)] = (uint32_t)i;
}

__global__ void cell_order_kernel(const int32_t* __restrict__ cell_rng, int64_t ncells, uint32_t* __restrict__ idx_sorted) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncells) return;
  const int s = cell_rng[2 * c], e = cell_rng[2 * c + 1];
  const int n = e - s;
  uint32_t* a = idx_sorted + s;
  if (n <= 16) {                      // the usual case (a few points per cell): insertion sort
    for (int i = 1; i < n; ++i) {
      const uint32_t v = a[i];
      int j = i - 1;
      while (j >= 0 && a[j] > v) {
        a[j + 1] = a[j];
        --j;
      }
      a[j + 1] = v;
    }
    return;
  }
  // a long run (thousands of points in one cell are possible: the caller chooses the cells): heapsort, O(n log n) whatever the
  // arrival order -- the insertion sort took minutes on a 75 000-element run
  auto sift = [&](int root, int end) {
    const uint32_t v = a[root];
    for (;;) {
      int child = 2 * root + 1;
      if (child >= end) break;
      if (child + 1 < end && a[child + 1] > a[child]) ++child;
      if (a[child] <= v) break;
      a[root] = a[child];
      root = child;
    }
    a[root] = v;
  };
  for (int i = n / 2 - 1; i >= 0; --i) sift(i, n);
  for (int end = n - 1; end > 0; --end) {
    const uint32_t t = a[0];
    a[0] = a[end];
    a[end] = t;
    sift(0, end);
  }
}

}  // namespace

extern "C" int64_t df_cell_sort_ws_bytes(int64_t ncells) {
  return ((ncells + CS_CHUNK - 1) / CS_CHUNK) * (int64_t)sizeof(int32_t);
}

// idx_sorted [n] u32, cell_rng [ncells][2] i32 (both fully written here), ws >= df_cell_sort_ws_bytes(ncells).
// Keys >= ncells are dropped (the caller's "invalid" sentinel).  Replaces the library radix sort of rounds 1-2.
extern "C" int df_cell_sort(const uint32_t* key, int64_t n, int64_t ncells, uint32_t* idx_sorted, int32_t* cell_rng, void* ws,
                            void* stream) {
  DF_REQUIRE(key && idx_sorted && cell_rng && ws && n > 0 && ncells > 0 && ncells < 0x3fffffffll && n < 0x7fffffffll, DF_E_ARG);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int nb = (int)((ncells + CS_CHUNK - 1) / CS_CHUNK);
  int32_t* blk = reinterpret_cast<int32_t*>(ws);
  hipError_t e = hipMemsetAsync(cell_rng, 0, (size_t)ncells * 2 * sizeof(int32_t), s);
  if (e != hipSuccess) return (int)e;
  const dim3 pts_grid((unsigned)((n + 255) / 256));
  hipLaunchKernelGGL(cell_hist_kernel, pts_grid, dim3(256), 0, s, key, n, (uint32_t)ncells, cell_rng);
  DF_CHECK_LAUNCH();
  hipLaunchKernelGGL(cell_blocksum_kernel, dim3(nb), dim3(256), 0, s, cell_rng, ncells, blk);
  DF_CHECK_LAUNCH();
  hipLaunchKernelGGL(cell_blockscan_kernel, dim3(1), dim3(256), 0, s, blk, nb);
  DF_CHECK_LAUNCH();
  hipLaunchKernelGGL(cell_offsets_kernel, dim3(nb), dim3(256), 0, s, cell_rng, ncells, blk);
  DF_CHECK_LAUNCH();
  hipLaunchKernelGGL(cell_scatter_kernel, pts_grid, dim3(256), 0, s, key, n, (uint32_t)ncells, cell_rng, idx_sorted);
  DF_CHECK_LAUNCH();
  hipLaunchKernelGGL(cell_order_kernel, dim3((unsigned)((ncells + 255) / 256)), dim3(256), 0, s, cell_rng, ncells, idx_sorted);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

extern "C" int df_pfn_bn_finalize(float* partial, int B, int nblk_stat, const int32_t* counts,
                                  const float* gamma, const float* beta, float eps, float momentum,
                                  float* running_mean, float* running_var, float* bn_ss, void* stream) {
  df_pillar_geom g = {};
  return df_pfn_bn_finalize2(partial, B, nblk_stat, counts, gamma, beta, eps, momentum, running_mean, running_var, bn_ss, nullptr, g, nullptr,
                             stream);
}

// ... also leaving an a-priori bound of max |canvas| in *canvas_bound (zero-initialised or holding another cloud's bound: integer
// atomic max) from the feature net's weights w_pfn [32][9] and the geometry -- see pfn_bn_finalize_kernel
extern "C" int df_pfn_bn_finalize2(float* partial, int B, int nblk_stat, const int32_t* counts,
                                   const float* gamma, const float* beta, float eps, float momentum,
                                   float* running_mean, float* running_var, float* bn_ss, const float* w_pfn, df_pillar_geom g,
                                   float* canvas_bound, void* stream) {
  DF_REQUIRE(partial && counts && bn_ss && B > 0 && nblk_stat > 0 && (!canvas_bound || w_pfn), DF_E_ARG);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(pfn_bn_finalize_kernel, dim3(B), dim3(1024), 0, s, partial, nblk_stat, counts, gamma, beta, eps, bn_ss, w_pfn, g,
                     reinterpret_cast<unsigned*>(canvas_bound));
  DF_CHECK_LAUNCH();
  if (running_mean && running_var) {
    hipLaunchKernelGGL(pfn_bn_running_kernel, dim3(1), dim3(32), 0, s, partial, B, nblk_stat, counts, momentum, running_mean,
                       running_var);
    DF_CHECK_LAUNCH();
  }
  return DF_OK;
}

extern "C" int df_pfn_bwd_stats(const float* pts_sorted, const int32_t* cell_rng, const uint32_t* key_sorted,
    const int32_t* counts, int B,
                                df_pillar_geom g, const float* w_pfn, const float* bn_ss, int bn_sample_stride,
                                int mode, df_img gout, float* partial, int nblk_stat, void* stream) {
  DF_REQUIRE(pts_sorted && cell_rng && key_sorted && counts && w_pfn && bn_ss && gout.ptr && partial && nblk_stat > 0 && geom_ok(g),
             DF_E_ARG);
  DF_REQUIRE(mode == 0 || mode == 1, DF_E_ARG);
  DF_REQUIRE(gout.n == B && gout.h == g.gy && gout.w == g.gx && gout.c == 32 && (gout.ld % 4) == 0, DF_E_SHAPE);
  const auto kern = mode == 0 ? pfn_bwd_stats_kernel<0> : pfn_bwd_stats_kernel<1>;
  hipLaunchKernelGGL(kern, dim3(nblk_stat, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), pts_sorted, cell_rng,
                     key_sorted, counts, g, w_pfn, bn_ss, bn_sample_stride, gout, partial);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

extern "C" int df_pfn_bwd_finalize(const float* partial, int B, int nblk_stat, const int32_t* counts, float* dgamma,
                                   float* dbeta, int accumulate, float* coef, void* stream) {
  DF_REQUIRE(partial && counts && dgamma && dbeta && coef && B > 0 && nblk_stat > 0, DF_E_ARG);
  hipLaunchKernelGGL(pfn_bwd_finalize_kernel, dim3(1), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), partial, B,
                     nblk_stat, counts, dgamma, dbeta, accumulate, coef);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

extern "C" int df_pfn_bwd_weights(const float* pts_sorted, const int32_t* cell_rng, const uint32_t* key_sorted,
    const int32_t* counts, int B,
                                  df_pillar_geom g, const float* w_pfn, const float* bn_ss, int bn_sample_stride,
                                  int mode, const float* coef, df_img gout, float* dw_partial, int nblk_stat,
                                  void* stream) {
  DF_REQUIRE(pts_sorted && cell_rng && key_sorted && counts && w_pfn && bn_ss && coef && gout.ptr && dw_partial && nblk_stat > 0 &&
                 geom_ok(g),
             DF_E_ARG);
  DF_REQUIRE(mode == 0 || mode == 1, DF_E_ARG);
  DF_REQUIRE(gout.n == B && gout.h == g.gy && gout.w == g.gx && gout.c == 32 && (gout.ld % 4) == 0, DF_E_SHAPE);
  const auto kern = mode == 0 ? pfn_bwd_weights_kernel<0> : pfn_bwd_weights_kernel<1>;
  hipLaunchKernelGGL(kern, dim3(nblk_stat, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     pts_sorted, cell_rng, key_sorted, counts, g, w_pfn, bn_ss, bn_sample_stride, coef, gout, dw_partial);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

extern "C" int df_pillar_input_grad(const uint32_t* key_sorted, const int32_t* counts, int B, int H, int W, int cloud,
                                    const float* dy1, const float* w1, df_img dskip, const float* w3, df_img dcanvas,
                                    int accumulate, int nblk, void* stream) {
  DF_REQUIRE(key_sorted && counts && dy1 && w1 && dskip.ptr && w3 && dcanvas.ptr && B > 0 && nblk > 0 && (cloud == 0 || cloud == 1),
             DF_E_ARG);
  DF_REQUIRE((H % 2) == 0 && (W % 2) == 0 && dskip.n == B && dskip.h == H && dskip.w == W && dskip.c == 64 && (dskip.ld % 2) == 0 &&
                 dcanvas.n == B && dcanvas.h == H && dcanvas.w == W && dcanvas.c == 32 &&
                 (int64_t)H * W * dskip.ld < (int64_t)0x30000000,   // 32-bit byte offsets inside one sample (buffer loads)
             DF_E_SHAPE);
  PillarGradParams p;
  p.key_sorted = key_sorted; p.counts = counts; p.B = B; p.H = H; p.W = W; p.cloud = cloud; p.accumulate = accumulate;
  p.dy1 = dy1; p.w1 = w1; p.dskip = dskip; p.w3 = w3; p.dcanvas = dcanvas;
  // DF_PIG_X3=0: the fp32-MFMA form of rounds 3-5 (also taken when the skip gradient's rows are not 16-byte aligned)
  static const bool x3 = !(getenv("DF_PIG_X3") && atoi(getenv("DF_PIG_X3")) == 0);
  if (x3 && (dskip.ld % 4) == 0 && (dskip.img_stride % 4) == 0 && df_aligned16(dskip.ptr) && df_aligned16(dy1) && W < 65536 &&
      (int64_t)H * W * dcanvas.ld < (int64_t)0x30000000) {
    const size_t lds3 = (size_t)(PG3_WT + (PG_THREADS / 64) * 4 * PGQ) * sizeof(float);
    DF_SET_LDS_ONCE((pillar_input_grad_x3_kernel), (int)lds3);
    hipLaunchKernelGGL(pillar_input_grad_x3_kernel, dim3(nblk, B), dim3(PG_THREADS), lds3, reinterpret_cast<hipStream_t>(stream), p);
    DF_CHECK_LAUNCH();
    return DF_OK;
  }
  const size_t lds_bytes = (size_t)(9 * 64 * 32 + 64 * 32 + (PG_THREADS / 64) * 16) * sizeof(float);
  DF_SET_LDS_ONCE((pillar_input_grad_kernel), (int)lds_bytes);
  hipLaunchKernelGGL(pillar_input_grad_kernel, dim3(nblk, B), dim3(PG_THREADS), lds_bytes,
                     reinterpret_cast<hipStream_t>(stream), p);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

extern "C" int df_sparse_wgrad3x3(const uint32_t* key_sorted, const int32_t* counts, int B, df_img x, df_img dy, float* ws,
                                  float* bias_ws, int nblk, void* stream) {
  DF_REQUIRE(key_sorted && counts && x.ptr && dy.ptr && ws && B > 0 && nblk > 0, DF_E_ARG);
  DF_REQUIRE(x.n == B && dy.n == B && x.c == 64 && dy.c == 64 && x.h == dy.h && x.w == dy.w && x.grp_size == x.n &&
                 dy.grp_size == dy.n,
             DF_E_SHAPE);
  DF_REQUIRE((int64_t)dy.h * dy.w * dy.ld < (int64_t)0x30000000 && (int64_t)x.h * x.w * x.ld < (int64_t)0x30000000, DF_E_SHAPE);   // 32-bit byte offsets
  SparseWgradParams p;
  p.key_sorted = key_sorted; p.counts = counts; p.H = dy.h; p.W = dy.w; p.dy = dy; p.x = x; p.ws = ws; p.bias_ws = bias_ws;
  hipLaunchKernelGGL(sparse_wgrad3x3_kernel, dim3(nblk, B), dim3(576), 0, reinterpret_cast<hipStream_t>(stream), p);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

// the bf16x2 form (sparse_wgrad3x3_x2_kernel: both operands as two bf16 planes, three v_mfma_f32_16x16x32_bf16 per tile and 32 pixels);
// same arguments, same partial layout; x / dy rows are read 16 bytes at a time
extern "C" int df_sparse_wgrad3x3_x2(const uint32_t* key_sorted, const int32_t* counts, int B, df_img x, df_img dy, float* ws,
                                     float* bias_ws, int nblk, void* stream) {
  DF_REQUIRE(key_sorted && counts && x.ptr && dy.ptr && ws && B > 0 && nblk > 0, DF_E_ARG);
  DF_REQUIRE(x.n == B && dy.n == B && x.c == 64 && dy.c == 64 && x.h == dy.h && x.w == dy.w && x.grp_size == x.n &&
                 dy.grp_size == dy.n && x.elt == 0 && dy.elt == 0,
             DF_E_SHAPE);
  DF_REQUIRE(df_aligned16(x.ptr) && df_aligned16(dy.ptr) && (x.ld % 4) == 0 && (dy.ld % 4) == 0 && (x.img_stride % 4) == 0 &&
                 (dy.img_stride % 4) == 0 && df_aligned16(ws) && (!bias_ws || df_aligned16(bias_ws)),
             DF_E_ALIGN);
  DF_REQUIRE((int64_t)dy.h * dy.w * dy.ld < (int64_t)0x30000000 && (int64_t)x.h * x.w * x.ld < (int64_t)0x30000000, DF_E_SHAPE);   // 32-bit byte offsets
  SparseWgradParams p;
  p.key_sorted = key_sorted; p.counts = counts; p.H = dy.h; p.W = dy.w; p.dy = dy; p.x = x; p.ws = ws; p.bias_ws = bias_ws;
  hipLaunchKernelGGL(sparse_wgrad3x3_x2_kernel, dim3(nblk, B), dim3(576), 0, reinterpret_cast<hipStream_t>(stream), p);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

extern "C" int df_sparse_conv3x3(const uint32_t* key_sorted, const int32_t* counts, int B, df_img x, const float* w,
                                 const float* bias, df_img y, int nblk, void* stream) {
  DF_REQUIRE(key_sorted && counts && x.ptr && y.ptr && w && B > 0 && nblk > 0, DF_E_ARG);
  DF_REQUIRE(x.n == B && y.n == B && x.c == 64 && y.c == 64 && x.h == y.h && x.w == y.w && (x.ld % 4) == 0 &&
                 df_aligned16(x.ptr) && x.grp_size == x.n && y.grp_size == y.n,
             DF_E_SHAPE);
  SparseConvParams p;
  p.key_sorted = key_sorted; p.counts = counts; p.H = y.h; p.W = y.w; p.x = x; p.y = y; p.w = w; p.bias = bias;
  const size_t lds_bytes = (size_t)(9 * 64 * 64 + (PG_THREADS / 64) * 16) * sizeof(float);
  DF_SET_LDS_ONCE((sparse_conv3x3_kernel), (int)lds_bytes);
  hipLaunchKernelGGL(sparse_conv3x3_kernel, dim3(nblk, B), dim3(PG_THREADS), lds_bytes,
                     reinterpret_cast<hipStream_t>(stream), p);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

// the fp16x2 form (sparse_conv3x3_h2_kernel): w2 = the [hi | lo] fp16 planes of w scaled by df_h2_scale(*w_amax) (df_split_h2 / the
// layer's planes from df_weight_prep), x_amax = an upper bound of max |x| (device scalars)
extern "C" int df_sparse_conv3x3_h2(const uint32_t* key_sorted, const int32_t* counts, int B, df_img x, const void* w2,
                                    const float* x_amax, const float* w_amax, const float* bias, df_img y, int nblk, void* stream) {
  DF_REQUIRE(key_sorted && counts && x.ptr && y.ptr && w2 && x_amax && w_amax && df_aligned16(w2) && B > 0 && nblk > 0, DF_E_ARG);
  DF_REQUIRE(x.n == B && y.n == B && x.c == 64 && y.c == 64 && x.h == y.h && x.w == y.w && (x.ld % 4) == 0 &&
                 df_aligned16(x.ptr) && x.grp_size == x.n && y.grp_size == y.n && x.elt == 0 && y.elt == 0,
             DF_E_SHAPE);
  SparseConvH2Params p;
  p.key_sorted = key_sorted; p.counts = counts; p.H = y.h; p.W = y.w; p.x = x; p.y = y; p.w2 = w2; p.bias = bias;
  p.amax_x = x_amax; p.amax_w = w_amax;
  const size_t lds_bytes = (size_t)(9 * 2 * 2 * 64 * 16 + (PG_THREADS / 64) * PGQ) * sizeof(float);
  DF_SET_LDS_ONCE((sparse_conv3x3_h2_kernel<false>), (int)lds_bytes);
  hipLaunchKernelGGL(sparse_conv3x3_h2_kernel<false>, dim3(nblk, B), dim3(PG_THREADS), lds_bytes,
                     reinterpret_cast<hipStream_t>(stream), p);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

// the bf16-operand form (mixed-precision training, Trainer(dtype="bf16")): x and w (fp32 [64,3,3,64]) rounded to bf16 on the way to the
// matrix pipe, fp32 accumulation -- what df_conv2d_w16 computes for the dense layers of that mode
extern "C" int df_sparse_conv3x3_bf16(const uint32_t* key_sorted, const int32_t* counts, int B, df_img x, const float* w,
                                      const float* bias, df_img y, int nblk, void* stream) {
  DF_REQUIRE(key_sorted && counts && x.ptr && y.ptr && w && df_aligned16(w) && B > 0 && nblk > 0, DF_E_ARG);
  DF_REQUIRE(x.n == B && y.n == B && x.c == 64 && y.c == 64 && x.h == y.h && x.w == y.w && (x.ld % 4) == 0 &&
                 df_aligned16(x.ptr) && x.grp_size == x.n && y.grp_size == y.n && x.elt == 0 && y.elt == 0,
             DF_E_SHAPE);
  SparseConvH2Params p;
  p.key_sorted = key_sorted; p.counts = counts; p.H = y.h; p.W = y.w; p.x = x; p.y = y; p.w2 = w; p.bias = bias;
  p.amax_x = nullptr; p.amax_w = nullptr;
  const size_t lds_bytes = (size_t)(9 * 2 * 2 * 64 * 16 + (PG_THREADS / 64) * PGQ) * sizeof(float);   // (the two-plane size: the queue sits behind it)
  DF_SET_LDS_ONCE((sparse_conv3x3_h2_kernel<true>), (int)lds_bytes);
  hipLaunchKernelGGL(sparse_conv3x3_h2_kernel<true>, dim3(nblk, B), dim3(PG_THREADS), lds_bytes,
                     reinterpret_cast<hipStream_t>(stream), p);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

extern "C" int df_sparse_in_wgrad(const uint32_t* key_sorted, const int32_t* counts, int B, int H, int W, int cloud,
                                  const float* dy1, df_img canvas, float* ws, int nblk, void* stream) {
  DF_REQUIRE(key_sorted && counts && dy1 && canvas.ptr && ws && B > 0 && nblk > 0 && (cloud == 0 || cloud == 1), DF_E_ARG);
  DF_REQUIRE((H % 2) == 0 && (W % 2) == 0 && canvas.n == B && canvas.h == H && canvas.w == W && canvas.c == 32, DF_E_SHAPE);
  DF_REQUIRE((int64_t)H * W * canvas.ld < (int64_t)0x30000000, DF_E_SHAPE);   // 32-bit byte offsets inside one sample
  DF_REQUIRE(df_aligned16(dy1) && (((uintptr_t)canvas.ptr) & 7) == 0 && (canvas.ld % 2) == 0 && (canvas.img_stride % 2) == 0 && W < 65536,
             DF_E_ALIGN);                                                    // 16-byte dy1 rows, 8-byte canvas pairs
  SparseInWgradParams p;
  p.key_sorted = key_sorted; p.counts = counts; p.B = B; p.H = H; p.W = W; p.cloud = cloud; p.dy1 = dy1; p.canvas = canvas;
  p.ws = ws;
  hipLaunchKernelGGL(sparse_in_wgrad_kernel, dim3(nblk, B), dim3(576), 0, reinterpret_cast<hipStream_t>(stream), p);
  DF_CHECK_LAUNCH();
  return DF_OK;
}
