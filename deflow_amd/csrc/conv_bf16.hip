// bf16 forward convolutions for the inference path of BASELINE configs[4] ("bf16 MFMA"): implicit GEMM on
// v_mfma_f32_32x32x16_bf16 (fp32 accumulation), NHWC bf16 activations, [Cout,kh,kw,Cin] bf16 weights.
//
// Same structure as conv_dma_kernel (conv.hip): 8-wave workgroups, 128 x BN tiles, operands by LDS-DMA into
// double-buffered unpadded tiles whose 16-byte slots are XOR-swizzled by (row >> 1) & 7, zero-fill of padding taps by
// out-of-range buffer offsets, (tap, k chunk) offsets in soffset.  A tile row is still 128 bytes -- now 64 bf16 of k --
// and one ds_read_b128 is exactly one MFMA operand (lane (row, kh) holds k = 16 s + 8 kh .. + 7 of k step s), so a stage
// is 4 MFMA k steps instead of 16: the matrix work per byte moved is 8x that of the fp32 kernel and the kernel is bound
// by the DMA / LDS path, not by the MFMA pipe.  Epilogues: bias, or bias + folded BatchNorm + exact-erf GELU (fp32 math),
// stored as bf16 or fp32.  Inference only (no statistics, no accumulate, no data-gradient mode).
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int BKH = 64;              // k elements per stage (128 bytes)
constexpr int LDB_ = 32;             // LDS row pitch in floats (128 bytes)
constexpr unsigned BAD16 = 0xFFFFFFFFu - (8u << 20);

struct ConvHParams {
  df_img x, y;             // element counts (c, ld, strides) are in bf16 / output elements
  const __bf16* w;
  const float* bias;
  const float* scale;
  const float* shift;
  int ks, stride, pad, epi, out_f32;
  int M, K, N, tiles_m, tiles_n, hw_y;
  unsigned x_bytes, w_bytes, dshift;
  unsigned y_bytes;        // extent of y in bytes (32-bit buffer offsets for the branch-free epilogue)
};

template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void epilogue_bf16(const ConvHParams& p, f32x16 (&acc)[BM / WM / 32][BN / WN / 32], float* lds, int m0,
                                              int n0) {
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, kh = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  // ---- epilogue: row -> output element offset table in LDS, then per-lane stores -----------------------------------
  int64_t* rowoff = reinterpret_cast<int64_t*>(lds);
  if (tid < BM) {
    const int m = m0 + tid;
    int64_t off = -1;
    if (m < p.M) {
      const int n = m / p.hw_y, rem = m - n * p.hw_y;
      off = df_img_base(p.y, n) + (int64_t)rem * p.y.ld;
    }
    rowoff[tid] = off;
  }
  __syncthreads();
  // straight-line buffer stores (rows past the end get an out-of-range offset and are dropped): a branch per element made
  // the compiler re-wait for the bias / scale loads -- and with them for the previous element's store -- 16 TM times
  constexpr unsigned ROW_BAD = 0xFFFFFFFFu - (8u << 20);
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y.ptr, 0, p.y_bytes, 0x00020000);
  const int esz = p.out_f32 ? 4 : 2;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int co = n0 + (wn * TN + j) * 32 + li;
    const float bia = p.bias ? p.bias[co] : 0.f;
    float sc = 1.f, sh = 0.f;
    if (p.epi == DF_EPI_BN_GELU) {
      sc = p.scale[co];
      sh = p.shift[co];
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      unsigned ob[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t off = rowoff[(wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh];
        ob[e] = off >= 0 ? (unsigned)((off + co) * esz) : ROW_BAD;
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float v = acc[i][j][e] + bia;
        if (p.epi == DF_EPI_BN_GELU) v = df_gelu(v * sc + sh);
        if (p.out_f32) {
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yr, ob[e], 0, 0);
        } else {
          const __bf16 h = (__bf16)v;
          __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, h), yr, ob[e], 0, 0);
        }
      }
    }
  }
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void conv_bf16_kernel(ConvHParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int NW = WM * WN, RP = 8 * NW;
  constexpr int RA = BM / RP, RB = (BN + RP - 1) / RP;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* As = lds;                      // [2][BM][32 floats = 64 bf16]
  float* Bs = lds + 2 * BM * LDB_;      // [2][BN][..]
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kh = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int swz = df_xcd_swizzle(blockIdx.x, gridDim.x);
  const int tile_n = swz % p.tiles_n, tile_m = swz / p.tiles_n;
  const int n0 = tile_n * BN, m0 = tile_m * BM;
  const int c4 = tid & 7, r0 = tid >> 3;
  const int hx = p.x.h, wx = p.x.w, ldx = p.x.ld;
  const int KC = p.K / BKH;

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<const char*>(p.x.ptr) - p.dshift), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
  const int c4s = c4 ^ ((r0 >> 1) & 7);
  unsigned aoff[RA], boff[RB];
  int ay[RA], ax[RA];
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    const int m = m0 + r0 + RP * i;
    ay[i] = ax[i] = -(1 << 28);
    aoff[i] = BAD16;
    if (m < p.M) {
      const int n = m / p.hw_y, rem = m - n * p.hw_y;
      const int oy = rem / p.y.w, ox = rem - oy * p.y.w;
      ay[i] = oy * p.stride - p.pad;
      ax[i] = ox * p.stride - p.pad;
      aoff[i] = (unsigned)((df_img_base(p.x, n) + ((int64_t)ay[i] * wx + ax[i]) * ldx + c4s * 8) * 2 + p.dshift);
    }
  }
#pragma unroll
  for (int i = 0; i < RB; ++i) boff[i] = (unsigned)(((int64_t)(n0 + r0 + RP * i) * p.ks * p.ks * p.K + c4s * 8) * 2);

  int l_kc = 0, l_iky = 0, l_ikx = 0;
  auto load_stage = [&](int buf) {
    const int ty = l_iky, tx = l_ikx;
    const unsigned soffA = (unsigned)(((ty * wx + tx) * ldx + l_kc * BKH) * 2);
    const unsigned soffB = (unsigned)(((ty * p.ks + tx) * p.K + l_kc * BKH) * 2);
    float* a = As + buf * BM * LDB_ + wave * 8 * LDB_;
    float* b = Bs + buf * BN * LDB_ + wave * 8 * LDB_;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const bool ok = (unsigned)(ay[i] + ty) < (unsigned)hx && (unsigned)(ax[i] + tx) < (unsigned)wx;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr_t)(a + i * RP * LDB_), 16, ok ? aoff[i] : BAD16, soffA, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < RB; ++i)
      if (BN % RP == 0 || wave * 8 + RP * i < BN)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_ptr_t)(b + i * RP * LDB_), 16, boff[i], soffB, 0, 0);
    if (++l_kc == KC) {
      l_kc = 0;
      if (++l_ikx == p.ks) {
        l_ikx = 0;
        ++l_iky;
      }
    }
  };
  int rslot[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) rslot[s] = ((2 * s + kh) ^ ((li >> 1) & 7)) * 4;   // float offset of the 16-byte slot

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nst = p.ks * p.ks * KC;
  load_stage(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int st = 0; st < nst; ++st) {
    const int buf = st & 1;
    if (st + 1 < nst) load_stage(buf ^ 1);
    const float* a = As + buf * BM * LDB_ + (wm * TM * 32 + li) * LDB_;
    const float* b = Bs + buf * BN * LDB_ + (wn * TN * 32 + li) * LDB_;
    f32x4 af[4][TM], bf[4][TN];   // all fragments of the stage first, then the MFMAs behind counted lgkmcnt waits
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int i = 0; i < TM; ++i) af[s][i] = ld4(a + i * 32 * LDB_ + rslot[s]);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[s][j] = ld4(b + j * 32 * LDB_ + rslot[s]);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[s][i]),
                                                              __builtin_bit_cast(bf16x8, bf[s][j]), acc[i][j], 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  epilogue_bf16<BM, BN, WM, WN>(p, acc, lds, m0, n0);
#endif
}

// Haloed-A form for 3x3 stride 1 (see conv_halo_kernel in conv.hip): one [130 px x 64 ch] A tile per (vertical tap, k chunk)
// serves the three horizontal taps -- for this DMA-bound kernel a third less traffic is worth proportionally more.
template <int BN, int WM, int WN>
__global__ __launch_bounds__(512) void conv_halo_bf16_kernel(ConvHParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 128, HR = 136;                 // halo tile rows: 130 used, padded to whole 8-row DMA instructions
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int RB = BN / 64;                       // B rows per thread (64 rows per DMA pass of 8 waves)
  static_assert(WM * WN == 8, "8 waves");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* As = lds;                      // [2][HR][128 bytes]
  float* Bs = lds + 2 * HR * LDB_;      // [2][BN][128 bytes]
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kh = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int swz = df_xcd_swizzle(blockIdx.x, gridDim.x);
  const int tile_n = swz % p.tiles_n, tile_m = swz / p.tiles_n;
  const int n0 = tile_n * BN, m0 = tile_m * BM;
  const int c4 = tid & 7, r0 = tid >> 3;            // DMA lane: physical slot, row within the 64-row pass
  const int hx = p.x.h, wx = p.x.w, ldx = p.x.ld;
  const int KC = p.K / BKH;
  const int n = m0 / p.hw_y, rem0 = m0 - n * p.hw_y;
  const int oy = rem0 / p.y.w, ox0 = rem0 - oy * p.y.w;

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<const char*>(p.x.ptr) - p.dshift), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
  // A halo: physical row j = 64 i + r0 (i = 0, 1, 2; rows >= 130 unused) holds input pixel (oy - 1 + ty, ox0 - 1 + j)
  unsigned aoff[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int j = 64 * i + r0;
    const int ix = ox0 - 1 + j;
    const bool ok = j < BM + 2 && ix >= 0 && ix < wx;
    const int slot = c4 ^ ((j >> 1) & 7);
    aoff[i] = ok ? (unsigned)((df_img_base(p.x, n) + ((int64_t)(oy - 1) * wx + ix) * ldx + slot * 8) * 2 + p.dshift) : BAD16;
  }
  const int c4b = c4 ^ ((r0 >> 1) & 7);
  unsigned boff[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) boff[i] = (unsigned)(((int64_t)(n0 + r0 + 64 * i) * 9 * p.K + c4b * 8) * 2);

  auto load_a = [&](int ty, int kc, int abuf) {     // A halo of group (ty, kc)
    const bool row_ok = (unsigned)(oy - 1 + ty) < (unsigned)hx;
    const unsigned soff = (unsigned)((ty * wx * ldx + kc * BKH) * 2);
    float* a = As + abuf * HR * LDB_ + wave * 8 * LDB_;
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (i < 2 || wave == 0)   // rows 128..135 ride on wave 0
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr_t)(a + i * 64 * LDB_), 16, row_ok ? aoff[i] : BAD16, soff, 0, 0);
  };
  auto load_b = [&](int ty, int tx, int kc, int bbuf) {
    const unsigned soff = (unsigned)(((ty * 3 + tx) * p.K + kc * BKH) * 2);
    float* b = Bs + bbuf * BN * LDB_ + wave * 8 * LDB_;
#pragma unroll
    for (int i = 0; i < RB; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_ptr_t)(b + i * 64 * LDB_), 16, boff[i], soff, 0, 0);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int ngroups = 3 * KC;
  load_a(0, 0, 0);
  load_b(0, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int st = 0;
  int ty = 0, kc = 0;   // group g = (ty, kc), counted without divisions (the scalar unit is shared by the CU's waves)
  for (int g = 0; g < ngroups; ++g) {
    const int kc1 = (kc + 1 == KC) ? 0 : kc + 1, ty1 = (kc + 1 == KC) ? ty + 1 : ty;   // group g + 1
#pragma unroll 1
    for (int tx = 0; tx < 3; ++tx, ++st) {
      // prefetch: the next stage's weights; at the first stage of a group also the next group's A halo
      if (tx < 2) load_b(ty, tx + 1, kc, (st + 1) & 1);
      else if (g + 1 < ngroups) load_b(ty1, 0, kc1, (st + 1) & 1);
      if (tx == 0 && g + 1 < ngroups) load_a(ty1, kc1, (g + 1) & 1);
      const float* a = As + (g & 1) * HR * LDB_ + (wm * TM * 32 + li + tx) * LDB_;
      const float* b = Bs + (st & 1) * BN * LDB_ + (wn * TN * 32 + li) * LDB_;
      const int sa = ((li + tx) >> 1) & 7, sb = (li >> 1) & 7;
      // all fragments of the stage first (4 k steps x (TM + TN) b128 reads), then the MFMAs behind counted lgkmcnt waits
      f32x4 af[4][TM], bf[4][TN];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int i = 0; i < TM; ++i) af[q][i] = ld4(a + i * 32 * LDB_ + (((2 * q + kh) ^ sa) * 4));
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[q][j] = ld4(b + j * 32 * LDB_ + (((2 * q + kh) ^ sb) * 4));
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[q][i]),
                                                                __builtin_bit_cast(bf16x8, bf[q][j]), acc[i][j], 0, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    ty = ty1;
    kc = kc1;
  }
  epilogue_bf16<BM, BN, WM, WN>(p, acc, lds, m0, n0);
#endif
}


template <int BN, int WM, int WN>
int launch_halo_bf16(const ConvHParams& p, hipStream_t s) {
  const size_t lds_bytes = (size_t)2 * (136 + BN) * LDB_ * sizeof(float);
  DF_SET_LDS_ONCE((conv_halo_bf16_kernel<BN, WM, WN>), (int)lds_bytes);
  hipLaunchKernelGGL((conv_halo_bf16_kernel<BN, WM, WN>), dim3(p.tiles_m * p.tiles_n), dim3(512), lds_bytes, s, p);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

template <int BM, int BN, int WM, int WN>
int launch_bf16(const ConvHParams& p, hipStream_t s) {
  const size_t lds_bytes = (size_t)2 * (BM + BN) * LDB_ * sizeof(float);
  DF_SET_LDS_ONCE((conv_bf16_kernel<BM, BN, WM, WN>), (int)lds_bytes);
  hipLaunchKernelGGL((conv_bf16_kernel<BM, BN, WM, WN>), dim3(p.tiles_m * p.tiles_n), dim3(64 * WM * WN), lds_bytes, s, p);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ x, __bf16* __restrict__ y, int64_t rows,
                                                         int cin, int ldx, int cout) {
  // y[row][c] = c < cin ? bf16(x[row * ldx + c]) : 0   (channel padding for the 32-channel network input);
  // 8 channels (two 16-byte loads, one 16-byte store) per thread; cin, cout, ldx multiples of 8
  const int C8 = cout >> 3;
  const int64_t total = rows * C8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / C8;
    const int c = (int)(i - r * C8) * 8;
    bf16x8 o;
    if (c < cin) {
      const f32x4 a = ld4(x + r * ldx + c), b = ld4(x + r * ldx + c + 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        o[k] = (__bf16)a[k];
        o[4 + k] = (__bf16)b[k];
      }
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = (__bf16)0.f;
    }
    *reinterpret_cast<bf16x8*>(y + i * 8) = o;
  }
}


// 3x3 stride-1 convolution for Cin = Cout = 64 as a ROLLING-ROW kernel: a persistent workgroup owns a strip of R output
// rows x 128 pixels of one image, keeps its share of the 9 x 64 x 64 weights in REGISTERS for the whole strip (144 VGPRs
// per lane: no weight traffic, no weight LDS reads) and streams the input rows through a 3-slot LDS ring: each haloed input
// row [130 px x 64 ch] arrives once and is multiplied into the three output rows it touches (vertical taps 0, 1, 2) --
// 36 MFMAs per wave per 17 KB of DMA, against 12 per group in conv_halo_bf16_kernel, no per-tile prologue, and a third of
// the input traffic.  Output row o is complete once input row o + 1 has been consumed; three accumulators rotate.
struct RollParams {
  ConvHParams c;
  int rows_per_wg, chunks, segs;
};

__global__ __launch_bounds__(512) void conv64_roll_bf16_kernel(RollParams rp) {
#if defined(__HIP_DEVICE_COMPILE__)
  const ConvHParams& p = rp.c;
  constexpr int HR = 136;
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [3][HR][128 bytes] + 8 wave-private output tiles
  float* stage = lds + 3 * HR * LDB_;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int id = blockIdx.x;
  const int chunk = id % rp.chunks, seg = (id / rp.chunks) % rp.segs, n = id / (rp.chunks * rp.segs);
  const int H = p.x.h, W = p.x.w, ldx = p.x.ld;
  const int y0 = chunk * rp.rows_per_wg, y1 = min(y0 + rp.rows_per_wg, H);
  const int ox0 = seg * 128;
  const int c4 = tid & 7, r0 = tid >> 3;

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<const char*>(p.x.ptr) - p.dshift), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y.ptr, 0, p.y_bytes, 0x00020000);
  // A halo: physical row j = 64 i + r0 holds input pixel x = ox0 - 1 + j of the current input row
  unsigned aoff[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int j = 64 * i + r0;
    const int ix = ox0 - 1 + j;
    const bool ok = j < 130 && ix >= 0 && ix < W;
    const int slot = c4 ^ ((j >> 1) & 7);
    // offsets are taken relative to row -1 (the base shift dshift = one row + one pixel keeps them >= 0)
    aoff[i] = ok ? (unsigned)((df_img_base(p.x, n) + ((int64_t)(-1) * W + ix) * ldx + slot * 8) * 2 + p.dshift) : BAD16;
  }
  auto load_row = [&](int r, int slot3) {           // input row r -> ring slot (zeros for rows outside the image)
    const bool row_ok = (unsigned)r < (unsigned)H;
    const unsigned soff = (unsigned)(((int64_t)(r + 1) * W * ldx) * 2);
    float* a = lds + slot3 * HR * LDB_ + wave * 8 * LDB_;
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (i < 2 || wave == 0)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr_t)(a + i * 64 * LDB_), 16, row_ok ? aoff[i] : BAD16, soff, 0, 0);
  };

  // this wave's weights: W[co = 32 wn + li][tap][k = 16 q + 8 kh .. + 7]
  bf16x8 wreg[9][4];
  {
    const __bf16* wp = p.w + (int64_t)(32 * wn + li) * 9 * 64 + 8 * kh;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) wreg[t][q] = *reinterpret_cast<const bf16x8*>(wp + t * 64 + 16 * q);
  }
  const int co = 32 * wn + li;
  const float bia = p.bias ? p.bias[co] : 0.f;
  float sc = 1.f, sh = 0.f;
  if (p.epi == DF_EPI_BN_GELU) {
    sc = p.scale[co];
    sh = p.shift[co];
  }
  constexpr unsigned ROW_BAD = 0xFFFFFFFFu - (8u << 20);
  const int64_t ybase = df_img_base(p.y, n);

  f32x16 acc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

  load_row(y0 - 1, 0);
  load_row(y0, 1);
  // one input row; PH (compile time) = ring slot of row r = accumulator of output row r - 1; it = r - (y0 - 1)
  auto step = [&](int r, int it, auto PH) {
    constexpr int S0 = decltype(PH)::value, S1 = (S0 + 1) % 3, S2 = (S0 + 2) % 3;
    // row r has landed once everything issued BEFORE the ops listed here is done: per earlier iteration 4 epilogue stores,
    // and the DMA of row r + 1 (2 instructions; 3 on wave 0) -- vmcnt retires in order
    const int newer = (wave == 0 ? 3 : 2) + 4 * min(it, 2);
    switch (newer) {
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
      case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    }
    // raw barrier: __syncthreads() would first drain vmcnt to 0 (the epilogue's global stores are pending behind its
    // release fence), i.e. also the prefetched row
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    load_row(r + 2, S2);      // slot of row r - 1, consumed before this barrier (rows past the strip are loaded but never used)
    const float* a = lds + S0 * HR * LDB_ + (wm * 32 + li) * LDB_;
    // input row r feeds output rows r - 1 (vertical tap 2), r (tap 1), r + 1 (tap 0) = accumulators S0, S1, S2
#pragma unroll
    for (int tx = 0; tx < 3; ++tx) {
      const int sa = ((li + tx) >> 1) & 7;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bf16x8 av = __builtin_bit_cast(bf16x8, ld4(a + tx * LDB_ + (((2 * q + kh) ^ sa) * 4)));
        acc[S0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, wreg[6 + tx][q], acc[S0], 0, 0, 0);
        acc[S1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, wreg[3 + tx][q], acc[S1], 0, 0, 0);
        acc[S2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, wreg[0 + tx][q], acc[S2], 0, 0, 0);
      }
    }
    // output row o = r - 1 is complete: bias / BN + GELU in registers, then through a wave-private LDS tile so that a lane
    // stores 16 contiguous bytes of one pixel (NST store instructions per row instead of 16 two- or four-byte ones).
    // Always NST stores (out of range when the row is not this strip's) so that the vmcnt bookkeeping above is uniform.
    const int o = r - 1;
    const bool row_valid = o >= y0 && o < y1;
    float* st = stage + wave * (32 * 36);            // [32 px][32 co] floats, pitch 36
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float v = acc[S0][e] + bia;
      if (p.epi == DF_EPI_BN_GELU) v = df_gelu(v * sc + sh);
      st[((e & 3) + 8 * (e >> 2) + 4 * kh) * 36 + li] = v;
      acc[S0][e] = 0.f;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (p.out_f32) {      // 32 px x 128 B: lane -> (px = lane >> 3 + 8 k, 4 channels)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int px = (lane >> 3) + 8 * k, c0 = (lane & 7) * 4;
        const f32x4 v = ld4(st + px * 36 + c0);
        const unsigned ob = row_valid ? (unsigned)((ybase + ((int64_t)o * W + ox0 + wm * 32 + px) * p.y.ld + 32 * wn + c0) * 4) : ROW_BAD;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yr, ob, 0, 0);
      }
    } else {              // 32 px x 64 B: lane -> (px = lane >> 2 + 16 k, 8 channels)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (k < 2) {
          const int px = (lane >> 2) + 16 * k, c0 = (lane & 3) * 8;
          const f32x4 a0 = ld4(st + px * 36 + c0), a1 = ld4(st + px * 36 + c0 + 4);
          bf16x8 h;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            h[u] = (__bf16)a0[u];
            h[4 + u] = (__bf16)a1[u];
          }
          const unsigned ob = row_valid ? (unsigned)((ybase + ((int64_t)o * W + ox0 + wm * 32 + px) * p.y.ld + 32 * wn + c0) * 2) : ROW_BAD;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, h), yr, ob, 0, 0);
        } else {          // two dummy stores: the same count as the fp32 branch
          __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, yr, ROW_BAD, 0, 0);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  for (int r = y0 - 1, it = 0; r <= y1; r += 3, it += 3) {
    step(r, it, std::integral_constant<int, 0>{});
    if (r + 1 > y1) break;
    step(r + 1, it + 1, std::integral_constant<int, 1>{});
    if (r + 2 > y1) break;
    step(r + 2, it + 2, std::integral_constant<int, 2>{});
  }
#endif
}

}  // namespace

extern "C" int df_conv2d_bf16(df_img x, const void* w, const float* bias, df_img y, int ksize, int stride, int pad, int epi,
                              const float* scale, const float* shift, int out_f32, void* stream) {
  DF_REQUIRE(x.ptr && y.ptr && w && df_aligned16(x.ptr) && df_aligned16(w), DF_E_ALIGN);
  DF_REQUIRE(x.n == y.n && (ksize == 1 || ksize == 3) && (stride == 1 || stride == 2) && pad == ksize / 2, DF_E_SHAPE);
  DF_REQUIRE(y.h == (x.h + 2 * pad - ksize) / stride + 1 && y.w == (x.w + 2 * pad - ksize) / stride + 1, DF_E_SHAPE);
  DF_REQUIRE((x.c % 64) == 0 && (y.c % 64) == 0 && (x.ld % 8) == 0 && x.grp_size > 0 && y.grp_size > 0, DF_E_SHAPE);
  DF_REQUIRE(epi == DF_EPI_BIAS || (epi == DF_EPI_BN_GELU && scale && shift), DF_E_ARG);
  ConvHParams p;
  p.x = x; p.y = y; p.w = reinterpret_cast<const __bf16*>(w); p.bias = bias; p.scale = scale; p.shift = shift;
  p.ks = ksize; p.stride = stride; p.pad = pad; p.epi = epi; p.out_f32 = out_f32;
  p.hw_y = y.h * y.w;
  const int64_t M = (int64_t)y.n * p.hw_y;
  DF_REQUIRE(M < (1ll << 31), DF_E_SHAPE);
  p.M = (int)M; p.K = x.c; p.N = y.c;
  const int bn = (p.N % 128 == 0) ? 128 : 64;
  p.tiles_m = (int)((M + 127) / 128);
  p.tiles_n = p.N / bn;
  const int64_t groups = x.n / x.grp_size;
  const int64_t ext = ((int64_t)(x.grp_size - 1) * x.img_stride + (groups - 1) * x.grp_off + (int64_t)x.h * x.w * x.ld) * 2;
  const int64_t dsh = (int64_t)pad * ((int64_t)x.w + 1) * x.ld * 2;
  const int64_t wb = (int64_t)p.N * ksize * ksize * p.K * 2;
  DF_REQUIRE(x.img_stride >= 0 && x.grp_off >= 0 && ext + dsh < (int64_t)BAD16 - (16 << 20) && wb < (1ll << 31), DF_E_SHAPE);
  p.x_bytes = (unsigned)(ext + dsh); p.w_bytes = (unsigned)wb; p.dshift = (unsigned)dsh;
  {
    const int64_t ygroups = y.n / y.grp_size, esz = out_f32 ? 4 : 2;
    const int64_t yext = ((int64_t)(y.grp_size - 1) * y.img_stride + (ygroups - 1) * y.grp_off + (int64_t)y.h * y.w * y.ld) * esz;
    DF_REQUIRE(y.img_stride >= 0 && y.grp_off >= 0 && yext < (int64_t)0xFFFFFFFFll - (16 << 20), DF_E_SHAPE);
    p.y_bytes = (unsigned)yext;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  static const int use_halo = getenv("DF_CONV_HALO") ? atoi(getenv("DF_CONV_HALO")) : 1;
  // small problems (B = 1 inference: the 64 x 64 and 128 x 128 levels): 64 x 64 tiles put four times the workgroups on
  // the 256 CUs instead of leaving half of them idle behind 128-row tiles.  DF_BF16_SMALL = tile-count threshold (0 = off)
  static const int small_tiles = getenv("DF_BF16_SMALL") ? atoi(getenv("DF_BF16_SMALL")) : 384;
  const bool small = (int64_t)p.tiles_m * p.tiles_n < small_tiles;
  static const int use_roll = getenv("DF_BF16_ROLL") ? atoi(getenv("DF_BF16_ROLL")) : 1;
  // rolling-row kernel: one workgroup per CU walking strips of rows -- needs enough rows per CU to amortise the 2 halo rows
  // and the weight load of a strip (B = 1 inference stays on the tiled kernel)
  static const int roll_min = getenv("DF_BF16_ROLL_MIN") ? atoi(getenv("DF_BF16_ROLL_MIN")) : 8192;
  const bool roll_ok = use_roll && (int64_t)y.n * (y.w / 128) * y.h >= roll_min;
  if (!small && roll_ok && ksize == 3 && stride == 1 && p.K == 64 && p.N == 64 && (y.w % 128) == 0 && x.w == y.w && x.h == y.h) {
    RollParams rp;
    rp.c = p;
    rp.segs = y.w / 128;
    // rows per workgroup: as long as possible (2 extra input rows per strip) while still >= 2 workgroups per CU
    int R = y.h;
    while (R > 16 && (int64_t)y.n * rp.segs * ((y.h + R - 1) / R) < 512) R = (R + 1) / 2;
    rp.rows_per_wg = R;
    rp.chunks = (y.h + R - 1) / R;
    const size_t lds_bytes = ((size_t)3 * 136 * LDB_ + 8 * 32 * 36) * sizeof(float);
    DF_SET_LDS_ONCE((conv64_roll_bf16_kernel), (int)lds_bytes);
    hipLaunchKernelGGL(conv64_roll_bf16_kernel, dim3((unsigned)(y.n * rp.segs * rp.chunks)), dim3(512), lds_bytes, s, rp);
    DF_CHECK_LAUNCH();
    return DF_OK;
  }
  if (!small && use_halo && ksize == 3 && stride == 1 && (y.w % 128) == 0 && x.w == y.w && x.h == y.h) {
    if (bn == 128) return launch_halo_bf16<128, 2, 4>(p, s);
    return launch_halo_bf16<64, 4, 2>(p, s);
  }
  if (small) {
    p.tiles_m = (int)((M + 63) / 64);
    p.tiles_n = p.N / 64;
    return launch_bf16<64, 64, 2, 2>(p, s);
  }
  if (bn == 128) return launch_bf16<128, 128, 2, 4>(p, s);
  return launch_bf16<128, 64, 4, 2>(p, s);
}

extern "C" int df_cast_bf16(const float* x, void* y, int64_t rows, int cin, int ldx, int cout, void* stream) {
  DF_REQUIRE(x && y && rows > 0 && cin > 0 && cout >= cin && ldx >= cin, DF_E_ARG);
  DF_REQUIRE((cin % 8) == 0 && (cout % 8) == 0 && (ldx % 4) == 0 && df_aligned16(x) && df_aligned16(y), DF_E_ALIGN);
  const int64_t total = rows * (cout / 8);
  const int64_t blocks = (total + 255) / 256;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, reinterpret_cast<__bf16*>(y), rows, cin, ldx, cout);
  DF_CHECK_LAUNCH();
  return DF_OK;
}
