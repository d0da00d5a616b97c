// BEV convolutions for the FastFlow3D UNet ([REF deflow.py:87-88]; block [REF decoder.py:202-220]).
//
// fp32 implicit GEMM on the CDNA4 matrix cores: v_mfma_f32_32x32x2_f32 (exact f32, 64 cyc / SIMD,
// 157 TFLOP/s chip peak).  Activations are NHWC so the reduction axis (input channels) is
// contiguous: a tile row is one 128-byte run of 32 channels of one (shifted) pixel.
//
//   conv_dma_kernel     y[m, co] = sum_{tap, ci} x[pix(m, tap), ci] * w[co, tap, ci]   (fwd and dgrad; operands by
//                       LDS-DMA; 8-wave workgroups for the 128 x 128 and 128 x 64 tiles)          -- the default
//   conv_kernel         the same with register-staged global -> LDS copies (tensors beyond 32-bit DMA offsets, A/B)
//   wgrad3_ring_kernel  dw[co, tap, ci] = sum_m dy[m, co] * x[pix(m, tap), ci], 3x3 stride 1, 12-wave workgroups
//   wgrad_kernel / wgrad1x1_kernel / wgrad_dma_kernel   the other kernel sizes / strides and the 4-wave DMA form
//
// LDS tiles are [row][32] floats, unpadded, with the 16-byte slots of a row XOR-swizzled by (row >> 1) & 7:
// ds_read_b128 fragments are conflict-free (each 16-lane service group hits 16 distinct (row parity, slot)
// pairs = all 64 banks; SQ_LDS_BANK_CONFLICT = 0 measured) and a 128x128 tile double-buffers in 64 KB, i.e. two
// workgroups per CU.  k is permuted inside each group of 8: MFMA step s of group g uses k = 8g + s on lanes 0-31
// and k = 8g + 4 + s on lanes 32-63 for BOTH operands, so one b128 read feeds four MFMA steps.
#include "conv_common.h"

namespace {


template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_kernel(ConvParams p) {
  constexpr int TM = BM / WM / 32;  // 32x32 MFMA tiles per wave along m
  constexpr int TN = BN / WN / 32;
  constexpr int RA = BM / 32;       // A rows staged per thread
  constexpr int RB = BN / 32;
  static_assert(WM * WN == 4, "4 waves");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* As = lds;                     // [2][BM][LDT]
  float* Bs = lds + 2 * BM * LDT;      // [2][BN][LDT]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, kh = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;

  const int swz = df_xcd_swizzle(blockIdx.x, gridDim.x);
  const int tile_n = swz % p.tiles_n, tile_m = swz / p.tiles_n;
  const int n0 = tile_n * BN;

  const float* __restrict__ xp = reinterpret_cast<const float*>(p.x.ptr);
  const int c4 = tid & 7, r0 = tid >> 3;
  const int hx = p.x.h, wx = p.x.w, ldx = p.x.ld;
  const int KC = p.K / BK;

  // tile geometry: plain row-major tiles, or one parity class of a stride-2 transposed conv
  RowDecode dec;
  dec.hw = p.hw_y; dec.w = p.y.w; dec.cls_mode = p.cls_tiles > 0; dec.py = dec.px = 0;
  dec.hh = p.y.h >> 1; dec.wh = p.y.w >> 1;
  int m0 = tile_m * BM, m_end = p.M;
  int ky0 = 0, kx0 = 0, kstep = 1, nky = p.ks, nkx = p.ks;
  if (dec.cls_mode) {
    const int cls = tile_m / p.cls_tiles;
    dec.py = cls >> 1; dec.px = cls & 1;
    m0 = (tile_m - cls * p.cls_tiles) * BM;
    m_end = p.y.n * dec.hh * dec.wh;
    kstep = 2;
    ky0 = (dec.py + p.pad) & 1; kx0 = (dec.px + p.pad) & 1;
    nky = (p.ks - ky0 + 1) >> 1; nkx = (p.ks - kx0 + 1) >> 1;
  }

  // per-thread description of the A rows it stages: row pointer (image base + this thread's 16-byte column),
  // and the top-left input coordinate.  Invalid rows point at element 0 and are masked at the LDS store.
  const float* arow[RA];
  int ay[RA], ax[RA];
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    const int m = m0 + r0 + 32 * i;
    arow[i] = xp + c4 * 4;
    ay[i] = ax[i] = -(1 << 28);  // never in range
    if (m < m_end) {
      int n, oy, ox;
      dec(m, n, oy, ox);
      arow[i] += df_img_base(p.x, n);
      if (p.mode == DF_CONV_FWD) {
        ay[i] = oy * p.stride - p.pad;
        ax[i] = ox * p.stride - p.pad;
      } else {
        ay[i] = oy + p.pad;
        ax[i] = ox + p.pad;
      }
    }
  }
  const float* brow[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) brow[i] = p.w + (int64_t)(n0 + r0 + 32 * i) * p.ks * p.ks * p.K + c4 * 4;

  // Stage loads are UNCONDITIONAL (clamped address, select at the LDS store) so the 2 * (RA + RB) global loads of a
  // stage issue back to back with no exec-mask branches or intermediate vmcnt waits.  (tap, k-chunk) advance
  // incrementally: load_stage is called for stages 0, 1, 2, ... in order.
  f32x4 areg[RA], breg[RB];
  unsigned amask = 0;
  int l_kc = 0, l_iky = 0, l_ikx = 0;
  const bool fwd = p.mode == DF_CONV_FWD;
  const bool half = !fwd && p.stride == 2;
  auto load_stage = [&]() {
    const int ky = ky0 + kstep * l_iky, kx = kx0 + kstep * l_ikx;
    const int koff = l_kc * BK;
    amask = 0;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      int iy = fwd ? ay[i] + ky : ay[i] - ky;
      int ix = fwd ? ax[i] + kx : ax[i] - kx;
      bool ok = true;
      if (half) {
        ok = !((iy | ix) & 1);
        iy >>= 1;
        ix >>= 1;
      }
      ok = ok && (unsigned)iy < (unsigned)hx && (unsigned)ix < (unsigned)wx;
      const int poff = ok ? (iy * wx + ix) * ldx : 0;
      areg[i] = ld4(arow[i] + poff + koff);
      amask |= (unsigned)ok << i;
    }
    const int woff = (ky * p.ks + kx) * p.K + koff;
#pragma unroll
    for (int i = 0; i < RB; ++i) breg[i] = ld4(brow[i] + woff);
    if (++l_kc == KC) {
      l_kc = 0;
      if (++l_ikx == nkx) {
        l_ikx = 0;
        ++l_iky;
      }
    }
  };
  const int wslot = (c4 ^ ((r0 >> 1) & 7)) * 4;   // swizzled slot of this thread's float4 (rows r0 + 32 i share it)
  int rslot[BK / 8];                              // swizzled slots of this lane's fragments
#pragma unroll
  for (int g = 0; g < BK / 8; ++g) rslot[g] = ((2 * g + kh) ^ ((li >> 1) & 7)) * 4;
  auto store_stage = [&](int buf) {
    float* a = As + buf * BM * LDT;
    float* b = Bs + buf * BN * LDT;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < RA; ++i) st4(a + (r0 + 32 * i) * LDT + wslot, ((amask >> i) & 1) ? areg[i] : zero);
#pragma unroll
    for (int i = 0; i < RB; ++i) st4(b + (r0 + 32 * i) * LDT + wslot, breg[i]);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nst = nky * nkx * KC;
  load_stage();
  store_stage(0);
  __syncthreads();
  for (int st = 0; st < nst; ++st) {
    const int buf = st & 1;
    if (st + 1 < nst && p.dbg == 0) load_stage();
    const float* a = As + buf * BM * LDT + (wm * TM * 32 + li) * LDT;
    const float* b = Bs + buf * BN * LDT + (wn * TN * 32 + li) * LDT;
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
      f32x4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = ld4(a + i * 32 * LDT + rslot[g]);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = ld4(b + j * 32 * LDT + rslot[g]);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
    }
    if (st + 1 < nst && p.dbg < 2) store_stage(buf ^ 1);
    __syncthreads();
  }

  conv_epilogue<BM, BN, WM, WN>(p, acc, lds, dec, m0, m_end, n0, tile_m);
}

// -------------------------------------------------------------------------------------------------------------
// LDS-DMA variant (the default): every stage's A and B tiles go HBM/L2 -> LDS with `buffer_load_dwordx4 ... lds`
// (no VGPR staging, no ds_write pass).  gfx950 semantics used (probed on hardware, tools/archive/dma_probe.hip): the wave's
// 64 lanes land at LDS base + lane * 16 (so one instruction fills 8 tile rows of 128 B); the per-lane SOURCE is free,
// so the XOR slot swizzle is applied to the source column; a lane whose voffset + soffset is outside the buffer
// writes ZEROS -- padding taps and out-of-tile rows cost one v_cndmask instead of a branch; the (tap, k-chunk)
// offset of a stage is wave-uniform and rides in soffset (SGPR), so per stage a thread only recomputes validity.

// H2 (fp32 training / inference, round 3; df_conv2d_h2f): fp32 tiles as in the fp32 form, every fragment split in REGISTERS into
// two scaled fp16 planes (df_h2_split: 22 significant bits) and multiplied as three v_mfma_f32_32x32x16_f16 (hi.hi' in one
// accumulator, the cross terms in a second one) instead of eight v_mfma_f32_32x32x2_f32 -- the 1x1 and stride-2 convolutions,
// whose im2col tiles have no halo to share, on the 16-bit matrix pipe.  The split costs ~6 VALU per element and fragment (the
// kernel becomes VALU-bound: ~1.8x the fp32-MFMA form on the stride-2 layers; the 1x1 layers are HBM-bound either way).
// BP (H2 only; round 4, second session): the WEIGHTS arrive pre-split (p.w2p: [hi plane | lo plane] fp16 of w s_w, the planes the
// 3x3 stride-1 kernels read, from the step's df_weight_prep launch): a weight row's stage is then 64 B of hi + 64 B of lo -- the
// same 128 bytes per row and stage, fetched by the same DMA instruction (lanes of slots 0-3 address the hi plane, 4-7 the lo plane)
// -- and a B fragment is two ds_read_b128 with no VALU.  The in-kernel split of B was redone by every wave that shares the
// fragment (4 of 8 waves on the 128 x 64 tile) at ~50 VALU instructions per 8 values: the kernel was VALU-bound 4 : 1 against
// its MFMAs on the 32 x 32 wave tiles.  A lane's 8 k values are CONSECUTIVE in this form (k = 16 q + 8 kh ..+7: the A slots
// change to match); same products, another order inside the MFMA's 16-deep sum.
// ---- UPS: the bilinear x2 half of an UpsampleSkip concatenation written by the skip convolution's workgroups (ConvParams::ups_t) ----
// PyTorch area_pixel_compute_source_index and the interpolation expression of elementwise.hip's upsample2x8_kernel (same values);
// eight channels per item, 16-byte plane stores
struct UpsLerp {
  int i0, i1;
  float l0, l1;
};
__device__ __forceinline__ UpsLerp ups_src(int dst, int in, int out, int align_corners) {
  float src;
  if (align_corners) {
    const float sc = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
    src = sc * dst;
  } else {
    src = ((float)dst + 0.5f) * ((float)in / (float)out) - 0.5f;
    if (src < 0.f) src = 0.f;
  }
  UpsLerp r;
  r.i0 = (int)src;
  if (r.i0 > in - 1) r.i0 = in - 1;
  r.i1 = r.i0 + (r.i0 < in - 1 ? 1 : 0);
  r.l1 = src - (float)r.i0;
  r.l0 = 1.f - r.l1;
  return r;
}
template <int BM, int BN, int NT>
__device__ __forceinline__ void ups_tile(const ConvParams& p, const RowDecode& dec, int m0, int m_end, int n0, int tid) {
  const float ys = df_h2_scale(*p.bound_y);
  const df_img& t = p.ups_t;
  const float* __restrict__ tp = reinterpret_cast<const float*>(t.ptr);
  char* y0 = reinterpret_cast<char*>(p.y.ptr) - (int64_t)t.c * 4;     // the concatenation's first half: t.c channels in front of y's
  constexpr int G = BN / 8;
  for (int it = tid; it < BM * G; it += NT) {
    const int r = it / G, c = n0 + (it - r * G) * 8;
    const int m = m0 + r;
    if (m >= m_end) continue;
    int n, Y, X;
    dec(m, n, Y, X);
    const UpsLerp ly = ups_src(Y, t.h, p.y.h, p.ups_ac), lx = ups_src(X, t.w, p.y.w, p.ups_ac);
    const float* b = tp + df_img_base(t, n) + c;
    const float* p00 = b + ((int64_t)ly.i0 * t.w + lx.i0) * t.ld;
    const float* p01 = b + ((int64_t)ly.i0 * t.w + lx.i1) * t.ld;
    const float* p10 = b + ((int64_t)ly.i1 * t.w + lx.i0) * t.ld;
    const float* p11 = b + ((int64_t)ly.i1 * t.w + lx.i1) * t.ld;
    const f32x4 a00 = ld4(p00), a01 = ld4(p01), a10 = ld4(p10), a11 = ld4(p11);
    const f32x4 b00 = ld4(p00 + 4), b01 = ld4(p01 + 4), b10 = ld4(p10 + 4), b11 = ld4(p11 + 4);
    const f32x4 o0 = ly.l0 * (lx.l0 * a00 + lx.l1 * a01) + ly.l1 * (lx.l0 * a10 + lx.l1 * a11);
    const f32x4 o1 = ly.l0 * (lx.l0 * b00 + lx.l1 * b01) + ly.l1 * (lx.l0 * b10 + lx.l1 * b11);
    f16x8_t hi, lo;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float t0 = o0[k] * ys, t1 = o1[k] * ys;
      hi[k] = (_Float16)t0;
      lo[k] = (_Float16)((t0 - (float)hi[k]) * H2_LO);
      hi[4 + k] = (_Float16)t1;
      lo[4 + k] = (_Float16)((t1 - (float)hi[4 + k]) * H2_LO);
    }
    const int64_t idx = df_img_base(p.y, n) + ((int64_t)Y * p.y.w + X) * p.y.ld + c;
    char* q = y0 + (idx & ~31ll) * 4 + (idx & 31) * 2;
    *reinterpret_cast<f16x8_t*>(q) = hi;
    *reinterpret_cast<f16x8_t*>(q + 64) = lo;
  }
}

template <int BM, int BN, int WM, int WN, bool BF = false, bool H2 = false, bool BP = false, bool UPS = false>
__global__ __launch_bounds__(64 * WM * WN) void conv_dma_kernel(ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)  // buffer-resource builtins only exist in the device pass; the host stub needs no body
  static_assert(!(BF && H2), "one operand format");
  static_assert(!BP || H2, "pre-split weights: the fp16x2 form only");
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int NW = WM * WN, RP = 8 * NW;       // waves; tile rows one DMA pass of the whole workgroup covers
  constexpr int RA = BM / RP, RB = BN / RP;
  static_assert(NW == 4 || NW == 8, "4 or 8 waves");  // 16 waves (32 x 32 wave tiles) measured 1-2 % slower than 8
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* As = lds;                     // [2][BM][LDT]
  float* Bs = lds + 2 * BM * LDT;      // [2][BN][LDT]
  typedef __attribute__((address_space(3))) void* lds_ptr_t;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kh = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;

  const int swz = df_xcd_swizzle(blockIdx.x, gridDim.x);
  const int tile_n = swz % p.tiles_n, tile_m = swz / p.tiles_n;
  const int n0 = tile_n * BN;
  const int c4 = tid & 7, r0 = tid >> 3;
  const int hx = p.x.h, wx = p.x.w, ldx = p.x.ld;
  const int KC = p.K / BK;

  RowDecode dec;
  dec.hw = p.hw_y; dec.w = p.y.w; dec.cls_mode = p.cls_tiles > 0; dec.py = dec.px = 0;
  dec.hh = p.y.h >> 1; dec.wh = p.y.w >> 1;
  int m0 = tile_m * BM, m_end = p.M;
  // tap enumeration: address taps (ty, tx) add (ty * wx + tx) pixels to the row's origin; weight tap index per mode
  int nky = p.ks, nkx = p.ks, ky0 = 0, kx0 = 0;
  const bool fwd = p.mode == DF_CONV_FWD;
  if (dec.cls_mode) {
    const int cls = p.cls_il ? (tile_m & 3) : tile_m / p.cls_tiles;
    dec.py = cls >> 1; dec.px = cls & 1;
    m0 = (p.cls_il ? (tile_m >> 2) : (tile_m - cls * p.cls_tiles)) * BM;
    m_end = p.y.n * dec.hh * dec.wh;
    ky0 = (dec.py + p.pad) & 1; kx0 = (dec.px + p.pad) & 1;
    nky = (p.ks - ky0 + 1) >> 1; nkx = (p.ks - kx0 + 1) >> 1;
  }

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<const char*>(p.x.ptr) - p.dshift), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)(BP ? p.w2p : (const void*)p.w), 0, p.w_bytes, 0x00020000);

  const int c4s = c4 ^ ((r0 >> 1) & 7);  // logical 16-B slot this lane fetches (it lands in physical slot c4)
  unsigned aoff[RA], boff[RB];
  int ay[RA], ax[RA];
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    const int m = m0 + r0 + RP * i;
    ay[i] = ax[i] = -(1 << 28);
    aoff[i] = DMA_BAD;
    if (m < m_end) {
      int n, oy, ox;
      dec(m, n, oy, ox);
      if (dec.cls_mode) {          // stride-2 transposed conv: input row = y' + dyt, dyt in {0, 1}
        ay[i] = oy >> 1;
        ax[i] = ox >> 1;
      } else if (fwd) {
        ay[i] = oy * p.stride - p.pad;
        ax[i] = ox * p.stride - p.pad;
      } else {                     // stride-1 dgrad = correlation with flipped taps
        ay[i] = oy - p.pad;
        ax[i] = ox - p.pad;
      }
      aoff[i] = (unsigned)((df_img_base(p.x, n) + ((int64_t)ay[i] * wx + ax[i]) * ldx + c4s * 4) * 4 + p.dshift);
    }
  }
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    if constexpr (BP)   // 2 bytes per element; logical slots 0-3 = hi k 0..31, 4-7 = lo k 0..31 (the lo plane follows the hi plane)
      boff[i] = (unsigned)((int64_t)(n0 + r0 + RP * i) * p.ks * p.ks * p.K * 2 + (c4s & 3) * 16 + (c4s >> 2) * (int64_t)(p.w_bytes / 2));
    else
      boff[i] = (unsigned)(((int64_t)(n0 + r0 + RP * i) * p.ks * p.ks * p.K + c4s * 4) * 4);
  }

  int l_kc = 0, l_iky = 0, l_ikx = 0;
  auto load_stage = [&](int buf) {
    int ty, tx, wtap;  // address tap, weight tap
    if (dec.cls_mode) {
      const int ky = ky0 + 2 * l_iky, kx = kx0 + 2 * l_ikx;
      ty = (dec.py + p.pad - ky) >> 1;
      tx = (dec.px + p.pad - kx) >> 1;
      wtap = ky * p.ks + kx;
    } else {
      ty = l_iky; tx = l_ikx;
      wtap = fwd ? ty * p.ks + tx : (p.ks - 1 - ty) * p.ks + (p.ks - 1 - tx);
    }
    const unsigned soffA = (unsigned)(((ty * wx + tx) * ldx + l_kc * BK) * 4);
    const unsigned soffB = (unsigned)((wtap * p.K + l_kc * BK) * (BP ? 2 : 4));
    float* a = As + buf * BM * LDT + wave * 8 * LDT;
    float* b = Bs + buf * BN * LDT + wave * 8 * LDT;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const bool ok = (unsigned)(ay[i] + ty) < (unsigned)hx && (unsigned)(ax[i] + tx) < (unsigned)wx;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr_t)(a + i * RP * LDT), 16, ok ? aoff[i] : DMA_BAD, soffA, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < RB; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_ptr_t)(b + i * RP * LDT), 16, boff[i], soffB, 0, 0);
    if (++l_kc == KC) {
      l_kc = 0;
      if (++l_ikx == nkx) {
        l_ikx = 0;
        ++l_iky;
      }
    }
  };
  int rslot[BK / 8];
#pragma unroll
  for (int g = 0; g < BK / 8; ++g) rslot[g] = ((2 * g + kh) ^ ((li >> 1) & 7)) * 4;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  f32x16 acc1[H2 ? TM : 1][H2 ? TN : 1];           // H2: the cross terms (hi lo' + lo hi'), scaled by 2048
  float sx = 1.f, sw = 1.f;
  if constexpr (H2) {
    sx = df_h2_scale(*p.amax_x);
    sw = df_h2_scale(*p.amax_w);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc1[i][j][e] = 0.f;
  }

  const int nst = nky * nkx * KC;
  load_stage(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int st = 0; st < nst; ++st) {
    const int buf = st & 1;
    if (st + 1 < nst) load_stage(buf ^ 1);   // buf^1 was last read in stage st-1; every wave has passed that barrier
    const float* a = As + buf * BM * LDT + (wm * TM * 32 + li) * LDT;
    const float* b = Bs + buf * BN * LDT + (wn * TN * 32 + li) * LDT;
    if constexpr (H2) {
#pragma unroll
      for (int q = 0; q < BK / 16; ++q) {
        f16x8_t ah[TM], al[TM], bh[TN], bl[TN];
        const int swz = (li >> 1) & 7;
        // BP: a lane's 8 k values are consecutive (fp32 slots 4 q + 2 kh, + 1) to match the pre-split weight planes
        const int sl0 = BP ? ((4 * q + 2 * kh) ^ swz) * 4 : rslot[2 * q], sl1 = BP ? ((4 * q + 2 * kh + 1) ^ swz) * 4 : rslot[2 * q + 1];
        auto split = [&](const float* row, float sc, f16x8_t& hi, f16x8_t& lo) {
          const f32x4 v0 = ld4(row + sl0), v1 = ld4(row + sl1);
          const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          df_h2_split(v, sc, hi, lo);
        };
#pragma unroll
        for (int i = 0; i < TM; ++i) split(a + i * 32 * LDT, sx, ah[i], al[i]);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if constexpr (BP) {   // hi: logical slot 2 q + kh, lo: 4 + 2 q + kh of the row's 128 bytes
            bh[j] = *reinterpret_cast<const f16x8_t*>(b + j * 32 * LDT + ((2 * q + kh) ^ swz) * 4);
            bl[j] = *reinterpret_cast<const f16x8_t*>(b + j * 32 * LDT + ((4 + 2 * q + kh) ^ swz) * 4);
          } else {
            split(b + j * 32 * LDT, sw, bh[j], bl[j]);
          }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc1[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
            acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc1[i][j], 0, 0, 0);
          }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      continue;
    }
    if constexpr (BF) {
      // bf16 operands: k groups (2 q, 2 q + 1) of a lane = 8 k values = one operand of v_mfma_f32_32x32x16_bf16
#pragma unroll
      for (int q = 0; q < BK / 16; ++q) {
        bf16x8_t a8[TM], b8[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
          a8[i] = pack_bf16(ld4(a + i * 32 * LDT + rslot[2 * q]), ld4(a + i * 32 * LDT + rslot[2 * q + 1]));
#pragma unroll
        for (int j = 0; j < TN; ++j)
          b8[j] = pack_bf16(ld4(b + j * 32 * LDT + rslot[2 * q]), ld4(b + j * 32 * LDT + rslot[2 * q + 1]));
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[i], b8[j], acc[i][j], 0, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      continue;
    }
    // fragments double-buffered in registers: the ds_read_b128s of k-group g+1 are issued before the MFMAs of group g
    f32x4 af[2][TM], bf[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) af[0][i] = ld4(a + i * 32 * LDT + rslot[0]);
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[0][j] = ld4(b + j * 32 * LDT + rslot[0]);
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
      if (g + 1 < BK / 8) {
#pragma unroll
        for (int i = 0; i < TM; ++i) af[(g + 1) & 1][i] = ld4(a + i * 32 * LDT + rslot[g + 1]);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[(g + 1) & 1][j] = ld4(b + j * 32 * LDT + rslot[g + 1]);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g & 1][i][s], bf[g & 1][j][s], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA for the next stage has landed
    __syncthreads();                                     // ... and everyone's
  }
  if constexpr (H2) {   // fold the cross terms in and take the two power-of-two scales out (exact multiplications)
    const float ix = 1.f / sx, iw = 1.f / sw;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = (acc[i][j][e] + acc1[i][j][e] * H2_LO_INV) * ix * iw;
  }
  conv_epilogue<BM, BN, WM, WN>(p, acc, lds, dec, m0, m_end, n0, tile_m);
  if constexpr (UPS) ups_tile<BM, BN, 64 * WM * WN>(p, dec, m0, m_end, n0, tid);
#endif
}

template <int BM, int BN, int WM, int WN>
int launch_conv(const ConvParams& p, hipStream_t s) {
  const size_t lds_bytes = (size_t)2 * (BM + BN) * LDT * sizeof(float);
  DF_SET_LDS_ONCE((conv_kernel<BM, BN, WM, WN>), (int)lds_bytes);
  DF_SET_LDS_ONCE((conv_dma_kernel<BM, BN, WM, WN>), (int)lds_bytes);
  if (p.x_bytes && p.bf16) {   // (the register-staged fallback for > 4 GB tensors stays fp32)
    DF_SET_LDS_ONCE((conv_dma_kernel<BM, BN, WM, WN, true>), (int)lds_bytes);
    hipLaunchKernelGGL((conv_dma_kernel<BM, BN, WM, WN, true>), dim3(p.tiles_m * p.tiles_n), dim3(256), lds_bytes, s, p);
  } else if (p.x_bytes)
    hipLaunchKernelGGL((conv_dma_kernel<BM, BN, WM, WN>), dim3(p.tiles_m * p.tiles_n), dim3(256), lds_bytes, s, p);
  else
    hipLaunchKernelGGL((conv_kernel<BM, BN, WM, WN>), dim3(p.tiles_m * p.tiles_n), dim3(256), lds_bytes, s, p);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

// 3x3 stride-1 convolution (forward, and data gradient = correlation with flipped taps) with a HALOED A tile: the 128
// output pixels of a tile are consecutive pixels of one image row (W % 128 == 0), so the A operands of the three
// horizontal taps are the same 130 input pixels shifted by one row of the LDS tile.  One A tile [130 x 32] per
// (vertical tap, k chunk) serves three stages; only the weight tile changes per stage: a third less DMA volume and
// DMA instructions than the per-tap im2col tiles of conv_dma_kernel (the in-loop DMA costs ~9 % there).  The XOR slot
// swizzle is keyed on the PHYSICAL tile row, and the b128 fragment reads stay conflict-free for any row shift (every
// 16-lane service group still touches 16 rows that are distinct mod 16).  8 waves, 2 workgroups per CU (66 KB LDS).
template <int BN, int WM, int WN, bool BF = false>
__global__ __launch_bounds__(512) void conv_halo_kernel(ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 128, HR = 136;                 // halo tile rows: 130 used, padded to whole 8-row DMA instructions
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int RB = BN / 64;                       // B rows per thread (64 rows per DMA pass of 8 waves)
  static_assert(WM * WN == 8, "8 waves");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* As = lds;                      // [2][HR][LDT]
  float* Bs = lds + 2 * HR * LDT;       // [2][BN][LDT]
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kh = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int swz = df_xcd_swizzle(blockIdx.x, gridDim.x);
  const int tile_n = swz % p.tiles_n, tile_m = swz / p.tiles_n;
  const int n0 = tile_n * BN, m0 = tile_m * BM;
  const int c4 = tid & 7, r0 = tid >> 3;            // DMA lane: physical slot, row within the 64-row pass
  const int hx = p.x.h, wx = p.x.w, ldx = p.x.ld;
  const int KC = p.K / BK;
  const bool fwd = p.mode == DF_CONV_FWD;
  RowDecode dec;
  dec.hw = p.hw_y; dec.w = p.y.w; dec.cls_mode = 0; dec.py = dec.px = 0; dec.hh = dec.wh = 0;
  int n, oy, ox0;
  dec(m0, n, oy, ox0);

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
    aoff[i] = ok ? (unsigned)((df_img_base(p.x, n) + ((int64_t)(oy - 1) * wx + ix) * ldx + slot * 4) * 4 + p.dshift) : DMA_BAD;
  }
  const int c4b = c4 ^ ((r0 >> 1) & 7);
  unsigned boff[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) boff[i] = (unsigned)(((int64_t)(n0 + r0 + 64 * i) * 9 * p.K + c4b * 4) * 4);

  auto load_a = [&](int ty, int kc, int abuf) {     // A halo of group (ty, kc)
    const bool row_ok = (unsigned)(oy - 1 + ty) < (unsigned)hx;
    const unsigned soff = (unsigned)((ty * wx * ldx + kc * BK) * 4);
    float* a = As + abuf * HR * LDT + wave * 8 * LDT;
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (i < 2 || wave == 0)   // rows 128..135 ride on wave 0
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr_t)(a + i * 64 * LDT), 16, row_ok ? aoff[i] : DMA_BAD, soff, 0, 0);
  };
  auto load_b = [&](int ty, int tx, int kc, int bbuf) {
    const int wtap = fwd ? ty * 3 + tx : (2 - ty) * 3 + (2 - tx);
    const unsigned soff = (unsigned)((wtap * p.K + kc * BK) * 4);
    float* b = Bs + bbuf * BN * LDT + wave * 8 * LDT;
#pragma unroll
    for (int i = 0; i < RB; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_ptr_t)(b + i * 64 * LDT), 16, boff[i], soff, 0, 0);
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
      const float* a = As + (g & 1) * HR * LDT + (wm * TM * 32 + li + tx) * LDT;
      const float* b = Bs + (st & 1) * BN * LDT + (wn * TN * 32 + li) * LDT;
      const int sa = ((li + tx) >> 1) & 7, sb = (li >> 1) & 7;
      if constexpr (BF) {
#pragma unroll
        for (int q = 0; q < BK / 16; ++q) {
          bf16x8_t a8[TM], b8[TN];
#pragma unroll
          for (int i = 0; i < TM; ++i)
            a8[i] = pack_bf16(ld4(a + i * 32 * LDT + (((4 * q + kh) ^ sa) * 4)), ld4(a + i * 32 * LDT + (((4 * q + 2 + kh) ^ sa) * 4)));
#pragma unroll
          for (int j = 0; j < TN; ++j)
            b8[j] = pack_bf16(ld4(b + j * 32 * LDT + (((4 * q + kh) ^ sb) * 4)), ld4(b + j * 32 * LDT + (((4 * q + 2 + kh) ^ sb) * 4)));
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[i], b8[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        continue;
      }
      f32x4 af[2][TM], bf[2][TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[0][i] = ld4(a + i * 32 * LDT + ((kh ^ sa) * 4));
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[0][j] = ld4(b + j * 32 * LDT + ((kh ^ sb) * 4));
#pragma unroll
      for (int q = 0; q < BK / 8; ++q) {
        if (q + 1 < BK / 8) {
#pragma unroll
          for (int i = 0; i < TM; ++i) af[(q + 1) & 1][i] = ld4(a + i * 32 * LDT + (((2 * (q + 1) + kh) ^ sa) * 4));
#pragma unroll
          for (int j = 0; j < TN; ++j) bf[(q + 1) & 1][j] = ld4(b + j * 32 * LDT + (((2 * (q + 1) + kh) ^ sb) * 4));
        }
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q & 1][i][s2], bf[q & 1][j][s2], acc[i][j], 0, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    ty = ty1;
    kc = kc1;
  }
  conv_epilogue<BM, BN, WM, WN>(p, acc, lds, dec, m0, p.M, n0, tile_m);
#endif
}

template <int BN, int WM, int WN>
static int launch_conv_halo(const ConvParams& p, hipStream_t s) {
  const size_t lds_bytes = (size_t)2 * (136 + BN) * LDT * sizeof(float);
  if (p.bf16) {
    DF_SET_LDS_ONCE((conv_halo_kernel<BN, WM, WN, true>), (int)lds_bytes);
    hipLaunchKernelGGL((conv_halo_kernel<BN, WM, WN, true>), dim3(p.tiles_m * p.tiles_n), dim3(512), lds_bytes, s, p);
    DF_CHECK_LAUNCH();
    return DF_OK;
  }
  DF_SET_LDS_ONCE((conv_halo_kernel<BN, WM, WN>), (int)lds_bytes);
  hipLaunchKernelGGL((conv_halo_kernel<BN, WM, WN>), dim3(p.tiles_m * p.tiles_n), dim3(512), lds_bytes, s, p);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

// ---- bf16-operand form of the haloed 3x3 kernel with bf16 TILES IN LDS (df_conv2d_w16) ------------------------------
// conv_halo_kernel<.., BF = true> keeps fp32 tiles in LDS and rounds the fragments as they leave it: every MFMA operand
// costs two ds_read_b128 + four v_cvt_pk, each element is converted once per consuming wave, and at 3 fragments per 2 MFMAs
// the kernel is LDS-bound at ~27 % matrix-pipe time (measured 22 %, profiles/r02_pmc_conv_bf16.txt).  Here
//   * the weights arrive pre-cast (bf16 copy of the [Cout, 9, K] tensor, one cast per step): LDS-DMA as before, half the bytes;
//   * the fp32 activation halo goes global -> registers -> v_cvt_pk -> LDS once per workgroup (8 floats = one 16-B slot
//     per thread and (ty, kc) group), prefetched one group ahead;
//   * every fragment is ONE ds_read_b128 (8 consecutive k of a row), no conversion in the loop.
// Rows are 64 B (32 bf16); 16-B slots XOR-swizzled with (row >> 2) & 3: any 16 consecutive rows (a b128 service group,
// whatever the tap shift) cover all 64 banks once.  Accumulator layout = conv_halo_kernel's, so the fp32 epilogues
// (bias / statistics / BN+GELU / accumulate) are shared.
// SEG = 1: the tile's 128 output pixels are one run of an image row (W % 128 == 0), halo = 130 pixels.  SEG = 2: W == 64, the
// tile is two whole image rows, halo = 2 x 66 pixels (halo row 66 s + c + tx feeds output pixel (oy + s, c)); 32-row MFMA
// blocks never straddle the two rows.
// X16: the activations are bfloat16 in memory (bf16-storage training): an item is ONE 16-byte load and goes to LDS as it is.
template <int BN, int WM, int WN, int SEG = 1, bool X16 = false>
__global__ __launch_bounds__(64 * WM * WN) void conv_halo_w16_kernel(ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 128, HR = 132;                 // halo rows: 130 (SEG = 1) / 132 (SEG = 2) used
  constexpr int SW = BM / SEG + 2;                  // halo pixels per segment
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int NW = WM * WN, NT = 64 * NW;         // waves, threads (8 waves: wave tile 64 x 32 at BN = 128; 4 waves: 64 x 64)
  constexpr int RB = (BN + 16 * NW - 1) / (16 * NW);   // weight DMA passes (a wave moves 16 rows of 64 B per instruction)
  constexpr int NIT = (SEG * SW * 4 + NT - 1) / NT;    // A staging items (one 16-byte LDS slot each) per thread
  constexpr int XES = X16 ? 2 : 4;                     // bytes per activation element in memory
  static_assert(NW == 8 || NW == 4, "4 or 8 waves");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* As = lds;                      // [2][HR][LDH]
  float* Bs = lds + 2 * HR * LDH;       // [2][3 taps][BN][LDH]
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kh = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int swz = df_xcd_swizzle(blockIdx.x, gridDim.x);
  const int tile_n = swz % p.tiles_n, tile_m = swz / p.tiles_n;
  const int n0 = tile_n * BN, m0 = tile_m * BM;
  const int hx = p.x.h, wx = p.x.w, ldx = p.x.ld;
  const int KC = p.K / BK;
  const bool fwd = p.mode == DF_CONV_FWD;
  RowDecode dec;
  dec.hw = p.hw_y; dec.w = p.y.w; dec.cls_mode = 0; dec.py = dec.px = 0; dec.hh = dec.wh = 0;
  int n, oy, ox0;
  dec(m0, n, oy, ox0);

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<const char*>(p.x.ptr) - p.dshift), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
  // A staging: item e of a thread = LDS slot (tid + e NT): halo row (tid + e NT) >> 2, physical slot tid & 3 (NT % 4 == 0).
  // Halo row j holds input pixel (oy - 1 + ty [+ segment], ox0 - 1 + j [- segment start]); physical slot s holds
  // k = 8 (s ^ swz(j)) .. + 7 of the chunk.  The last item covers the few rows past a multiple of NT / 4.
  const int aslot = tid & 3;
  unsigned aoff[NIT];
  int aseg[NIT];
  bool aon[NIT];
#pragma unroll
  for (int e = 0; e < NIT; ++e) {
    const int j = (tid + e * NT) >> 2;
    const int sg = SEG == 1 ? 0 : j / SW;
    const int ix = ox0 - 1 + j - sg * SW;
    aon[e] = j < SEG * SW;
    aseg[e] = sg;
    const int sl = aslot ^ ((j >> 2) & 3);
    aoff[e] = (aon[e] && ix >= 0 && ix < wx)
                  ? (unsigned)((df_img_base(p.x, n) + ((int64_t)(oy + sg - 1) * wx + ix) * ldx + sl * 8) * XES + p.dshift) : DMA_BAD;
  }
  const int brow = wave * 16 + (lane >> 2), bslot = (lane & 3) ^ ((lane >> 4) & 3);   // (row >> 2) & 3 = (lane >> 4) & 3
  unsigned boff[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) boff[i] = (unsigned)(((int64_t)(n0 + brow + 16 * NW * i) * 9 * p.K + bslot * 8) * 2);

  // Scalar-light group addressing (profiles/r02_pmc_conv_bf16.txt: 994 k SALU vs 745 k VALU instructions per launch; the
  // scalar unit is shared by the CU's 16 waves and 12 MFMAs per wave and group are only ~400 cycles): the per-group scalar
  // offsets are CARRIED and advanced by adds -- no multiplies, no forward / data-gradient tap selects inside the loop.
  //   sa  = byte offset of group (ty, kc)'s halo row / k chunk in x:  += BK * 4 per chunk, += row pitch - K * 4 per tap row
  //   swb = byte offset of tap (ty, tx = 0), chunk kc in the bf16 weights; tap tx sits tx * dtap further (dtap < 0 for the
  //         data gradient, whose taps run backwards: 8 - 3 ty - tx)
  const unsigned a_row_step = (unsigned)(wx * ldx * XES - KC * BK * XES);
  const int dtap = fwd ? p.K * 2 : -p.K * 2;
  const int w_row_step = 3 * dtap - KC * BK * 2;
  unsigned sa_next = 0;                               // offsets of the NEXT group to fetch (group 0 first)
  int swb_next = fwd ? 0 : 8 * p.K * 2;
  int ty_next = 0;
  f32x4 ra[NIT][2];
  auto fetch_a = [&]() {                            // the next group's halo -> registers
    const unsigned soff = sa_next;
    const int ty = ty_next;
#pragma unroll
    for (int e = 0; e < NIT; ++e) {
      if (e + 1 < NIT || aon[e]) {
        const unsigned v = (unsigned)(oy + aseg[e] - 1 + ty) < (unsigned)hx ? aoff[e] : DMA_BAD;
        ra[e][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, v, soff, 0));
        if constexpr (!X16) ra[e][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, v + 16, soff, 0));
      }
    }
  };
  auto stash_a = [&](int abuf) {                    // registers -> bf16 -> LDS
    float* a = As + abuf * HR * LDH;
#pragma unroll
    for (int e = 0; e < NIT; ++e)
      if (e + 1 < NIT || aon[e]) {
        if constexpr (X16) *reinterpret_cast<f32x4*>(a + (tid + e * NT) * 4) = ra[e][0];
        else *reinterpret_cast<bf16x8_t*>(a + (tid + e * NT) * 4) = pack_bf16(ra[e][0], ra[e][1]);
      }
  };
  auto load_b = [&](int bbuf) {                     // the three horizontal taps' weight tiles of the next group
#pragma unroll
    for (int tx = 0; tx < 3; ++tx) {
      const unsigned soff = (unsigned)(swb_next + tx * dtap);
      float* b = Bs + (bbuf * 3 + tx) * BN * LDH + wave * 16 * LDH;
#pragma unroll
      for (int i = 0; i < RB; ++i)
        if (wave * 16 + 16 * NW * i < BN)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_ptr_t)(b + i * 16 * NW * LDH), 16, boff[i], soff, 0, 0);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // One stage = one (vertical tap, k chunk) group = three taps x two k-steps (12 MFMAs per wave at 128 x 128): a third of the
  // barriers of the per-tap pipeline, and the next group's operands have a whole group of MFMAs to arrive.
  const int ngroups = 3 * KC;
  int kc_next = 0;
  auto advance = [&]() {                            // (ty, kc) of the next group to fetch and its carried offsets
    if (++kc_next == KC) {
      kc_next = 0;
      ++ty_next;
      sa_next += a_row_step + BK * XES;
      swb_next += w_row_step + BK * 2;
    } else {
      sa_next += BK * XES;
      swb_next += BK * 2;
    }
  };
  fetch_a();
  load_b(0);
  advance();
  stash_a(0);
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), as a builtin: visible to the compiler's wait-count bookkeeping
  __syncthreads();
  const float* a_cur = As + (wm * TM * 32 + li) * LDH;      // this group's tiles; the two ring slots alternate
  const float* b_cur = Bs + (wn * TN * 32 + li) * LDH;
  int a_flip = HR * LDH, b_flip = 3 * BN * LDH;
  for (int g = 0; g < ngroups; ++g) {
    if (g + 1 < ngroups) {
      load_b((g + 1) & 1);
      fetch_a();
      advance();
    }
    const float* a0 = a_cur;
    const float* b0 = b_cur;
    a_cur += a_flip; a_flip = -a_flip;
    b_cur += b_flip; b_flip = -b_flip;
    const int sb = (li >> 2) & 3;
#pragma unroll
    for (int tx = 0; tx < 3; ++tx) {
      const float* a = a0 + tx * LDH;
      const float* b = b0 + tx * BN * LDH;
#pragma unroll
      for (int q = 0; q < BK / 16; ++q) {
        bf16x8_t a8[TM], b8[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          // output rows of block i sit 2 halo rows further per completed segment (SEG = 2: blocks 2, 3 of the tile)
          const int sh = SEG == 1 ? 0 : 2 * ((wm * TM + i) * 32 / (BM / SEG));
          const int sa = ((li + tx + sh) >> 2) & 3;
          a8[i] = *reinterpret_cast<const bf16x8_t*>(a + (i * 32 + sh) * LDH + (((2 * q + kh) ^ sa) * 4));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) b8[j] = *reinterpret_cast<const bf16x8_t*>(b + j * 32 * LDH + (((2 * q + kh) ^ sb) * 4));
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[i], b8[j], acc[i][j], 0, 0, 0);
      }
    }
    // the next group's halo: its ring slot was last read in group g - 1 (behind the previous barrier)
    if (g + 1 < ngroups) stash_a((g + 1) & 1);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), as a builtin: visible to the compiler's wait-count bookkeeping
    __syncthreads();
  }
  conv_epilogue<BM, BN, WM, WN>(p, acc, lds, dec, m0, p.M, n0, tile_m);
#endif
}

template <int BN, int WM, int WN, int SEG = 1, bool X16 = false>
static int launch_conv_halo_w16(const ConvParams& p, hipStream_t s) {
  const size_t tiles = (size_t)2 * (132 + 3 * BN) * LDH * sizeof(float);
  const size_t epi = (size_t)(2 * 128 + WM * BN * 2) * sizeof(float);   // conv_epilogue: rowoff[BM] (int64) + red[WM][BN][2]
  const size_t lds_bytes = tiles > epi ? tiles : epi;
  DF_SET_LDS_ONCE((conv_halo_w16_kernel<BN, WM, WN, SEG, X16>), (int)lds_bytes);
  hipLaunchKernelGGL((conv_halo_w16_kernel<BN, WM, WN, SEG, X16>), dim3(p.tiles_m * p.tiles_n), dim3(64 * WM * WN), lds_bytes, s, p);
  DF_CHECK_LAUNCH();
  return DF_OK;
}


// ---- fp32-ACCURATE 3x3 stride-1 convolution on the bf16 matrix pipe ("bf16x3", df_conv2d_x3) ---------------------------
// The fp32 MFMA (v_mfma_f32_32x32x2_f32) moves 2 k per 64 cycles; v_mfma_f32_32x32x16_bf16 moves 16 k per 32: sixteen times the
// k throughput.  An fp32 value is EXACTLY the sum of three bf16 values (8 + 8 + 8 mantissa bits: hi = bf16(x), mid = bf16(x -
// hi), lo = bf16(x - hi - mid)), and a bf16 x bf16 product is exact in the fp32 accumulator, so
//     x w  =  xh wh + xh wm + xm wh + xm wm + xh wl + xl wh  + (terms of relative size <= 2^-24: dropped)
// -- six bf16 MFMAs reproduce the fp32 product to fp32 rounding (the same accumulation as before), at 16 / 6 = 2.7x the fp32
// MFMA rate.  Round 2 tried the split on the FRAGMENTS inside conv_halo_kernel (7 VALU instructions per MFMA: no gain).  Here
// the split happens ONCE per tile element: weights are pre-split into three bf16 planes per optimizer step (df_split_bf16x3),
// the activation halo is split in the global -> register -> LDS staging of the bf16-tile kernel (once per workgroup and (ty, kc)
// group, ~10 VALU per element against 4608 MFMA flops it feeds), and the loop is ds_read_b128 + MFMA only: 18 fragment reads
// per 24 MFMAs and wave.  Six times the matrix work per byte also takes the kernel off the latency bound of the bf16-tile
// form (12 MFMAs per wave between barriers, shorter than its own prefetch): a stage here is one TAP -- 24 MFMAs, ~770 matrix
// cycles per wave -- and the weight tiles run through a DB-deep LDS ring with COUNTED vmcnt (DB - 1 stages = ~2 us of
// prefetch), the halo is fetched a whole group (three stages) ahead.  LDS: A [2][3 planes][132][64 B] = 50 KB + B
// [DB][3][BN][64 B] = 96 KB: one 8-wave workgroup per CU.
// BM = 128, SEG = 1: the tile is 128 pixels of one image row (W % 128 == 0); BM = 128, SEG = 2: two whole rows of a W == 64
// image (halo 2 x 66); BM = 256, SEG = 1: 256 pixels of one row (W % 256 == 0) -- the form of the 64-output-channel layers, whose
// 128 x 64 tile had 32 x 32 wave tiles (12 MFMAs per 12 fragment reads); with 256 rows the wave tile is 64 x 32 again.
// NP = 2 ("fp16x2", df_conv2d_h2): TWO fp16 planes per operand instead of three bf16 ones.  With per-tensor power-of-two scales
// (df_h2_scale of an upper bound of max|x| resp. max|w|; the caller measures them, df_absmax) x s = hi + lo / 2048 carries 22
// significant bits, and  x w = [hi hi' + (hi lo' + lo hi') / 2048] / (s s')  + O(2^-22 |x w|): THREE fp16 MFMAs (the hi.hi terms
// in one accumulator, the two cross terms in a second one that is folded in at the end) instead of six -- half the matrix
// work of the bf16x3 form, 2 / 3 of its LDS bytes, at an error 3 x 2^-22 per product, i.e. far below the fp32 accumulation's own
// rounding over K = 9 Cin terms.
// NP = 1: ONE bf16 plane -- the bf16-operand training mode (df_conv2d_w16) on this kernel's large tiles; X16: the activations are
// bfloat16 in memory (bf16-storage training): one 16-byte load per staging item, stored to LDS as it is.
// XP (round 4, NP = 2 only; df_conv2d_h2 with x.elt = 2): the activations arrive PRE-SPLIT -- x is an "h2" image (per pixel and
// 32-channel chunk one 128-byte line [32 fp16 hi | 32 fp16 lo] of x s, s = df_h2_scale(*amax_x), written by the producer with
// the bound amax_x known before it wrote) -- so the halo goes global -> LDS by DMA like the weights: no staging registers, no
// split in VALU (3.8 VALU per MFMA in the fp32-input form), no LDS stores.  One op = 16 halo rows x 64 B of one plane; every
// wave issues the same NAO ops per group (spare slots repeat the first ops: same bytes to the same place) so that the counted
// waits stay compile-time constants.
template <int BM, int BN, int WM, int WN, int SEG, int DB, int NP = 3, bool X16 = false, bool XP = false, bool BWS = false>
__global__ __launch_bounds__(64 * WM * WN) void conv_halo_x3_kernel(ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int SW = BM / SEG + 2;                     // halo pixels per segment
  constexpr int HR = XP ? (SEG * SW + 15) / 16 * 16 : (SEG * SW + 3) / 4 * 4;   // halo rows (130 / 132 / 258 used; XP: whole 16-row DMA ops)
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int NW = WM * WN, NT = 64 * NW;
  constexpr int NIT = XP ? 1 : (SEG * SW * 4 + NT - 1) / NT;    // A staging items (one 16-byte bf16 slot = 8 floats) per thread
  constexpr int NAOPS = NP * HR / 16;                  // XP: 1-KB DMA ops per halo (both planes)
  constexpr int NAO = (NAOPS + NW - 1) / NW;           // XP: ... per wave
  static_assert(!XP || (NP == 2 && !X16), "pre-split input: the fp16x2 form only");
  constexpr int AP = HR * LDH, AB = NP * AP;           // A plane / buffer (floats)
  constexpr int BP = BN * LDH, BSL = NP * BP;          // B plane / ring slot (floats)
  constexpr int PD = DB - 1;                           // prefetch distance of the weight ring (stages)
  static_assert(NW == 8 && (BN == 128 || BN == 64) && DB >= 3 && (BM == 128 || BM == 256 || BM == 512) && (SEG == 1 || SEG == 2 || SEG == 4) &&
                (NP >= 1 && NP <= 3) && (!X16 || NP == 1), "8 waves");
  constexpr int XE = X16 ? 2 : 4;                       // bytes per activation element in memory
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* As = lds;                      // [2][NP][HR][LDH]
  float* Bs = lds + 2 * AB;             // [DB][NP][BN][LDH]
  float sx = 1.f, sw = 1.f;
  if constexpr (NP == 2) { sx = df_h2_scale(*p.amax_x); sw = df_h2_scale(*p.amax_w); }
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kh = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int swz = df_xcd_swizzle(blockIdx.x, gridDim.x);
  const int tile_n = swz % p.tiles_n, tile_m = swz / p.tiles_n;
  const int n0 = tile_n * BN, m0 = tile_m * BM;
  const int hx = p.x.h, wx = p.x.w, ldx = p.x.ld;
  const int KC = p.K / BK;
  const bool fwd = p.mode == DF_CONV_FWD;
  RowDecode dec;
  dec.hw = p.hw_y; dec.w = p.y.w; dec.cls_mode = 0; dec.py = dec.px = 0; dec.hh = dec.wh = 0;
  int n, oy, ox0;
  dec(m0, n, oy, ox0);

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<const char*>(p.x.ptr) - p.dshift), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
  // A staging (as conv_halo_w16_kernel): item e = LDS slot (tid + e NT): halo row j = slot >> 2, physical 16-byte slot tid & 3
  // holding k = 8 (s ^ ((j >> 2) & 3)) .. + 7 of the chunk.  Halo row j = segment sg, pixel j - sg SW: input pixel
  // (oy + sg - 1 + ty, ox0 - 1 + j - sg SW).  EVERY thread issues every item's two loads (out-of-range items read zeros from an
  // out-of-range offset): the counted vmcnt waits below need the same number of operations in every wave.
  const int aslot = tid & 3;
  unsigned aoff[NIT];
  int aseg[NIT];
  bool aon[NIT];
#pragma unroll
  for (int e = 0; e < NIT; ++e) {
    const int j = (tid + e * NT) >> 2;
    const int sg = SEG == 1 ? 0 : j / SW;
    const int ix = ox0 - 1 + j - sg * SW;
    aon[e] = j < SEG * SW;
    aseg[e] = sg;
    const int sl = aslot ^ ((j >> 2) & 3);
    aoff[e] = (aon[e] && ix >= 0 && ix < wx)
                  ? (unsigned)((df_img_base(p.x, n) + ((int64_t)(oy + sg - 1) * wx + ix) * ldx + sl * 8) * XE + p.dshift) : DMA_BAD;
  }
  // weight DMA: a wave moves 16 rows x 64 B of one plane per instruction; plane pl sits pl * N * 9 * K elements further.
  // BN = 64: waves 4 .. 7 repeat the rows of waves 0 .. 3 (same data to the same place) so that EVERY wave issues exactly three
  // operations per weight stage -- the counted waits below are then compile-time constants, which the compiler's own
  // wait-count bookkeeping can follow (run-time counts made it drain vmcnt to 0 at every halo fetch)
  const int brow = ((wave * 16) % BN) + (lane >> 2), bslot = (lane & 3) ^ ((lane >> 4) & 3);
  const unsigned boff = (unsigned)(((int64_t)(n0 + brow) * 9 * p.K + bslot * 8) * 2);
  const unsigned plane_bytes = (unsigned)((int64_t)p.N * 9 * p.K * 2);
  constexpr int NBW = NP, NFA = XP ? NAO : (X16 ? 1 : 2) * NIT;  // VMEM operations per wave: per weight stage / per halo fetch
  // XP: op e of this wave = halo op (wave + e NW) mod NAOPS = plane pl, rows rb 16 .. + 15; lane = (row rb 16 + (lane >> 2),
  // physical slot lane & 3 <- logical 16-byte quarter (lane & 3) ^ ((row >> 2) & 3) of the chunk's 64-byte plane line)
  unsigned poff[XP ? NAO : 1];
  int pseg[XP ? NAO : 1], pdst[XP ? NAO : 1];
  if constexpr (XP) {
#pragma unroll
    for (int e = 0; e < NAO; ++e) {
      const int jo = (wave + e * NW) % NAOPS;
      const int pl = jo / (HR / 16), rb = jo - pl * (HR / 16);
      const int j = rb * 16 + (lane >> 2);
      const int sg = SEG == 1 ? 0 : min(j / SW, SEG - 1);
      const int ix = ox0 - 1 + j - sg * SW;
      const int sl = (lane & 3) ^ ((j >> 2) & 3);
      pseg[e] = sg;
      pdst[e] = pl * AP + rb * 16 * LDH;
      poff[e] = (j < SEG * SW && ix >= 0 && ix < wx)
                    ? (unsigned)((df_img_base(p.x, n) + ((int64_t)(oy + sg - 1) * wx + ix) * ldx) * 4 + pl * 64 + sl * 16 + p.dshift) : DMA_BAD;
    }
  }

  // carried scalar offsets (see conv_halo_w16_kernel): halo row / chunk of the NEXT group to fetch, tap (ty, tx, kc) of the next
  // weight stage to issue
  const unsigned a_row_step = (unsigned)(wx * ldx * XE - KC * BK * XE);
  const int dtap = fwd ? p.K * 2 : -p.K * 2;
  // ROTATED tap rows (round 4; p.rot): the workgroup walks ty = r, r + 1, r + 2 (mod 3) with r = (1 - oy) mod 3, so that in its
  // phase ph EVERY workgroup reads the input rows = ph (mod 3).  The 32 workgroups an XCD runs side by side hold 32 consecutive
  // rows of one image (df_xcd_swizzle); in the plain order row R was fetched by workgroup R + 1 in its first third, by R in its
  // second and by R - 1 in its last, with 32 x 131 KB of other rows through the 4 MB L2 in between: the 512-wide layers fetched
  // 2.65x their input from HBM (FETCH_SIZE, round 4).  Rotated, the three readers of a row ask for it in
  // the same phase.  (The fp32 accumulation order of the taps now depends on oy: deterministic, not the plain order's bits.)
  // Measured (FETCH_SIZE, profiles/r04_pmc_conv_rot.txt): 512-wide rows 2.84 -> 1.47 GB per launch (64 channels), 1.54 -> 0.86 GB
  // (128); two-row tiles -10..-25 %; four-row tiles read every row in all three phases either way (+15 %: left in the plain order).
  // The kernels' times do not move (+-1 %): they were not waiting for these bytes.
  const int rot = (p.rot && SEG <= 2) ? (1 + 2 * oy) % 3 : 0;
  const unsigned a_row = (unsigned)(wx * ldx * XE);
  unsigned sa_next = rot * a_row;
  int ty_next = rot, kc_next = 0;
  int swb_next = (fwd ? 0 : 8 * p.K * 2) + rot * 3 * dtap;   // byte offset of tap (ty, tx = 0), chunk kc of the next weight GROUP
  int btx = 0, bkc = 0, bty = rot;                     // next weight stage: its tap inside the group, the group's k chunk and tap row
  const int NG = 3 * KC, NS = 3 * NG;

  f32x4 ra[NIT][2];
  auto fetch_a = [&](int abuf) {                      // the next group's halo -> registers (XP: -> LDS buffer abuf by DMA), then advance the group cursor
    if constexpr (XP) {
      float* a = As + abuf * AB;
#pragma unroll
      for (int e = 0; e < NAO; ++e) {
        const unsigned v = (unsigned)(oy + pseg[e] - 1 + ty_next) < (unsigned)hx ? poff[e] : DMA_BAD;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr_t)(a + pdst[e]), 16, v, sa_next, 0, 0);
      }
    } else {
#pragma unroll
    for (int e = 0; e < NIT; ++e) {
      const unsigned v = (unsigned)(oy + aseg[e] - 1 + ty_next) < (unsigned)hx ? aoff[e] : DMA_BAD;
      ra[e][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, v, sa_next, 0));
      if constexpr (!X16) ra[e][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, v + 16, sa_next, 0));
    }
    }
    if (++kc_next == KC) {
      kc_next = 0;
      sa_next += a_row_step + BK * XE;
      if (++ty_next == 3) {                            // (rotated order: wrap to the first tap row)
        ty_next = 0;
        sa_next -= 3 * a_row;
      }
    } else {
      sa_next += BK * XE;
    }
  };
  auto stash_a = [&](int abuf) {                      // registers -> three bf16 (two fp16 / one bf16) planes -> LDS
    if constexpr (XP) return;                         // (the DMA wrote the planes)
    float* a = As + abuf * AB;
#pragma unroll
    for (int e = 0; e < NIT; ++e) {
      if (e + 1 < NIT || aon[e]) {
        if constexpr (X16) {                            // eight bfloat16 as loaded
          *reinterpret_cast<f32x4*>(a + (tid + e * NT) * 4) = ra[e][0];
          continue;
        }
        float v[8], r[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] = ra[e][0][k]; v[4 + k] = ra[e][1][k]; }
        if constexpr (NP == 1) {
          *reinterpret_cast<bf16x8_t*>(a + (tid + e * NT) * 4) = pack_bf16(v);
          continue;
        }
        if constexpr (NP == 2) {
          f16x8_t h, l;
          df_h2_split(v, sx, h, l);
          float* d = a + (tid + e * NT) * 4;
          *reinterpret_cast<f16x8_t*>(d) = h;
          *reinterpret_cast<f16x8_t*>(d + AP) = l;
          continue;
        }
        bf16x8_t hi, mi, lo;
#pragma unroll
        for (int k = 0; k < 8; ++k) { hi[k] = (__bf16)v[k]; r[k] = v[k] - (float)hi[k]; }
#pragma unroll
        for (int k = 0; k < 8; ++k) { mi[k] = (__bf16)r[k]; r[k] = r[k] - (float)mi[k]; }
#pragma unroll
        for (int k = 0; k < 8; ++k) lo[k] = (__bf16)r[k];
        float* d = a + (tid + e * NT) * 4;
        *reinterpret_cast<bf16x8_t*>(d) = hi;
        *reinterpret_cast<bf16x8_t*>(d + AP) = mi;
        *reinterpret_cast<bf16x8_t*>(d + 2 * AP) = lo;
      }
    }
  };
  // the next weight stage (tap btx of the group at swb_next): three plane tiles into the next ring slot
  int bslot_ring = 0;
  auto issue_b = [&]() {
    float* b = Bs + bslot_ring * BSL + ((wave * 16) % BN) * LDH;
    const unsigned soff = (unsigned)(swb_next + btx * dtap);
#pragma unroll
    for (int pl = 0; pl < NP; ++pl)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_ptr_t)(b + pl * BP), 16, boff, soff + pl * plane_bytes, 0, 0);
    if (++bslot_ring == DB) bslot_ring = 0;
    if (++btx == 3) {                                 // next weight group: next k chunk, or the next tap row's first
      btx = 0;
      if (++bkc == KC) {
        bkc = 0;
        swb_next += 3 * dtap - KC * BK * 2 + BK * 2;
        if (++bty == 3) {
          bty = 0;
          swb_next -= 9 * dtap;
        }
      } else {
        swb_next += BK * 2;
      }
    }
  };
#define DF_VMCNT(N) __builtin_amdgcn_s_waitcnt(0x0F70 | ((N) & 15) | (((N) >> 4) << 14))

  f32x16 acc[TM][TN];
  f32x16 acc1[NP == 2 ? TM : 1][NP == 2 ? TN : 1];     // fp16x2: the cross terms (hi lo' + lo hi'), scaled by 2048
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        acc[i][j][e] = 0.f;
        if constexpr (NP == 2) acc1[i][j][e] = 0.f;
      }

  int cur_slot = 0;
  // one tap stage: issue (MAIN: the next group's halo at tx == 0, weight stage s + PD), 2 k-steps x 6 products, the waits
  auto stage = [&](auto tx_c, auto main_c, int g, int s) {
    constexpr int tx = decltype(tx_c)::value;
    constexpr bool MAIN = decltype(main_c)::value;
    const float* a0 = As + (g & 1) * AB + (wm * TM * 32 + li) * LDH;
    if constexpr (MAIN) {
      if (tx == 0) fetch_a((g + 1) & 1);
      issue_b();
    } else {
      if (tx == 0 && g + 1 < NG) fetch_a((g + 1) & 1);
      if (s + PD < NS) issue_b();
    }
    const float* b0 = Bs + cur_slot * BSL + (wn * TN * 32 + li) * LDH;
    if (++cur_slot == DB) cur_slot = 0;
    const float* a = a0 + tx * LDH;
    const int sb = (li >> 2) & 3;
#pragma unroll
    for (int q = 0; q < BK / 16; ++q) {
      if constexpr (NP == 1) {
        bf16x8_t a1[TM], b1[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int sh = SEG == 1 ? 0 : 2 * ((wm * TM + i) * 32 / (BM / SEG));
          const int sa = ((li + tx + sh) >> 2) & 3;
          a1[i] = *reinterpret_cast<const bf16x8_t*>(a + (i * 32 + sh) * LDH + (((2 * q + kh) ^ sa) * 4));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) b1[j] = *reinterpret_cast<const bf16x8_t*>(b0 + j * 32 * LDH + (((2 * q + kh) ^ sb) * 4));
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[i], b1[j], acc[i][j], 0, 0, 0);
        continue;
      }
      if constexpr (NP == 2) {
        f16x8_t ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int sh = SEG == 1 ? 0 : 2 * ((wm * TM + i) * 32 / (BM / SEG));
          const int sa = ((li + tx + sh) >> 2) & 3;
          const float* ap = a + (i * 32 + sh) * LDH + (((2 * q + kh) ^ sa) * 4);
          ah[i] = *reinterpret_cast<const f16x8_t*>(ap);
          al[i] = *reinterpret_cast<const f16x8_t*>(ap + AP);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const float* bp = b0 + j * 32 * LDH + (((2 * q + kh) ^ sb) * 4);
          bh[j] = *reinterpret_cast<const f16x8_t*>(bp);
          bl[j] = *reinterpret_cast<const f16x8_t*>(bp + BP);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc1[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
            acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc1[i][j], 0, 0, 0);
          }
        continue;
      }
      bf16x8_t ah[TM], am[TM], al[TM], bh[TN], bm[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        // output rows of block i sit 2 halo rows further per completed segment (SEG = 2: blocks 2, 3 of the tile)
        const int sh = SEG == 1 ? 0 : 2 * ((wm * TM + i) * 32 / (BM / SEG));
        const int sa = ((li + tx + sh) >> 2) & 3;
        const float* ap = a + (i * 32 + sh) * LDH + (((2 * q + kh) ^ sa) * 4);
        ah[i] = *reinterpret_cast<const bf16x8_t*>(ap);
        am[i] = *reinterpret_cast<const bf16x8_t*>(ap + AP);
        al[i] = *reinterpret_cast<const bf16x8_t*>(ap + 2 * AP);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const float* bp = b0 + j * 32 * LDH + (((2 * q + kh) ^ sb) * 4);
        bh[j] = *reinterpret_cast<const bf16x8_t*>(bp);
        bm[j] = *reinterpret_cast<const bf16x8_t*>(bp + BP);
        bl[j] = *reinterpret_cast<const bf16x8_t*>(bp + 2 * BP);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          f32x16 c = acc[i][j];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], c, 0, 0, 0);   // small terms first
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bm[j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bh[j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bm[j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], c, 0, 0, 0);
          acc[i][j] = c;
        }
    }
    // ---- end of the stage: (last stage of a group) the next group's halo -> LDS; then weight stage s + 1 must have landed.
    // MAIN (every issue above happened): operations issued after the halo fetch = the 3 weight stages of this group; after
    // weight stage s + 1's = PD - 1 weight stages + the halo fetches of the PD - 1 stages behind it.
    if constexpr (MAIN) {
      if (tx == 2) {
        DF_VMCNT(3 * NBW);
        stash_a((g + 1) & 1);
      }
      // halo fetches among stages s - (PD - 2) .. s: taps tx, tx - 1, ... (mod 3) -- count the zeros
      constexpr int FA_IN = ((PD - 1) / 3) + (((PD - 1) % 3) > tx ? 1 : 0);
      DF_VMCNT((PD - 1) * NBW + FA_IN * NFA);
    } else {
      DF_VMCNT(0);
      if (tx == 2 && g + 1 < NG) stash_a((g + 1) & 1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };

  // prologue: halo of group 0, weight stages 0 .. PD - 1 (NS = 27 KC >= PD)
  fetch_a(0);
#pragma unroll
  for (int d = 0; d < PD; ++d) issue_b();
  DF_VMCNT(PD * NBW);                                  // the halo loads were issued first
  stash_a(0);
  DF_VMCNT((PD - 1) * NBW);                            // weight stage 0
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  // main groups: every stage s of the group has s + PD < NS, a next group to fetch, and (PD > 3) a full window of PD - 1
  // earlier stages behind it; the first group of a deep ring and the last groups take the conservative form
  const int g_main = max(min((NS - PD) / 3, NG - 1), 0);
  using T0 = std::integral_constant<int, 0>;
  using T1 = std::integral_constant<int, 1>;
  using T2 = std::integral_constant<int, 2>;
  int g = 0, s = 0;
  if (PD > 3) {
    stage(T0{}, std::false_type{}, g, s);
    stage(T1{}, std::false_type{}, g, s + 1);
    stage(T2{}, std::false_type{}, g, s + 2);
    ++g; s += 3;
  }
  for (; g < g_main; ++g, s += 3) {
    stage(T0{}, std::true_type{}, g, s);
    stage(T1{}, std::true_type{}, g, s + 1);
    stage(T2{}, std::true_type{}, g, s + 2);
  }
  for (; g < NG; ++g, s += 3) {                        // tail: run-time issue conditions, vmcnt(0) waits
    stage(T0{}, std::false_type{}, g, s);
    stage(T1{}, std::false_type{}, g, s + 1);
    stage(T2{}, std::false_type{}, g, s + 2);
  }
#undef DF_VMCNT
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  if constexpr (NP == 2) {   // fold the cross terms in and take the two power-of-two scales out (exact multiplications)
    const float ix = 1.f / sx, iw = 1.f / sw;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = (acc[i][j][e] + acc1[i][j][e] * H2_LO_INV) * ix * iw;
  }
  conv_epilogue<BM, BN, WM, WN, BWS>(p, acc, lds, dec, m0, p.M, n0, tile_m);
#endif
}

template <int BM, int BN, int WM, int WN, int SEG, int DB, int NP = 3, bool X16 = false, bool XP = false, bool BWS = false>
static int launch_conv_halo_x3(const ConvParams& p, hipStream_t s) {
  constexpr int HR = XP ? (SEG * (BM / SEG + 2) + 15) / 16 * 16 : (SEG * (BM / SEG + 2) + 3) / 4 * 4;
  const size_t lds_bytes = (size_t)(2 * NP * HR + DB * NP * BN) * LDH * sizeof(float);
  DF_SET_LDS_ONCE((conv_halo_x3_kernel<BM, BN, WM, WN, SEG, DB, NP, X16, XP, BWS>), (int)lds_bytes);
  hipLaunchKernelGGL((conv_halo_x3_kernel<BM, BN, WM, WN, SEG, DB, NP, X16, XP, BWS>), dim3(p.tiles_m * p.tiles_n), dim3(64 * WM * WN), lds_bytes, s, p);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

// w [n] fp32 -> two fp16 planes out2[2][n] of w s (s = df_h2_scale(*amax)): hi = fp16(w s), lo = fp16((w s - hi) 2048)
__global__ __launch_bounds__(256) void split_h2_kernel(const float* __restrict__ w, const float* __restrict__ amax,
                                                       _Float16* __restrict__ out2, int64_t n) {
  const float s = df_h2_scale(*amax);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float t = w[i] * s;
    const _Float16 hi = (_Float16)t;
    out2[i] = hi;
    out2[n + i] = (_Float16)((t - (float)hi) * H2_LO);
  }
}

// amax (a float's bit pattern, zero-initialised by the caller) <- max(amax, max |x|) over an image view: non-negative floats
// order like their bit patterns, so an integer atomic max is exact and order-independent.  One thread per 4 channels.
__global__ __launch_bounds__(256) void absmax_kernel(df_img d, int64_t total4, unsigned* __restrict__ amax) {
  const int c4 = d.c >> 2;
  const int64_t hw = (int64_t)d.h * d.w;
  float mf = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / c4;
    const int q = (int)(i - row * c4);
    const int64_t n = row / hw;
    const float* src = reinterpret_cast<const float*>(d.ptr) + df_img_base(d, (int)n) + (row - n * hw) * d.ld + q * 4;
    mf = df_amax4(mf, ld4(src));
  }
  df_block_amax(mf, amax);
}

// w [n] fp32 -> three bf16 planes out3[3][n]: hi = bf16(w), mid = bf16(w - hi), lo = bf16(w - hi - mid)  (w == hi + mid + lo)
__global__ __launch_bounds__(256) void split_bf16x3_kernel(const float* __restrict__ w, __bf16* __restrict__ out3, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = w[i];
    const __bf16 hi = (__bf16)v;
    const float r1 = v - (float)hi;
    const __bf16 mi = (__bf16)r1;
    const float r2 = r1 - (float)mi;
    out3[i] = hi;
    out3[n + i] = mi;
    out3[2 * n + i] = (__bf16)r2;
  }
}

// 8-wave forms (512 threads; wave tile 64 x 32 resp. 32 x 32, 32 / 16 accumulator registers): the same LDS footprint
// and DMA traffic as the 4-wave kernels but twice the waves per SIMD to cover each other's barrier and DMA waits
// (measured +2..4 % on the 128 x 128 tile).  DMA path only.
template <int BM, int BN, int WM, int WN>
static int launch_conv_w8(const ConvParams& p, hipStream_t s) {
  const size_t lds_bytes = (size_t)2 * (BM + BN) * LDT * sizeof(float);
  if (p.bf16) {
    DF_SET_LDS_ONCE((conv_dma_kernel<BM, BN, WM, WN, true>), (int)lds_bytes);
    hipLaunchKernelGGL((conv_dma_kernel<BM, BN, WM, WN, true>), dim3(p.tiles_m * p.tiles_n), dim3(64 * WM * WN), lds_bytes, s, p);
    DF_CHECK_LAUNCH();
    return DF_OK;
  }
  if (p.ups_on) {                        // ... and the bilinear half of the concatenation written by the same workgroups
    DF_REQUIRE(p.amax_x && p.amax_w && p.w2p, DF_E_ARG);
    DF_SET_LDS_ONCE((conv_dma_kernel<BM, BN, WM, WN, false, true, true, true>), (int)lds_bytes);
    hipLaunchKernelGGL((conv_dma_kernel<BM, BN, WM, WN, false, true, true, true>), dim3(p.tiles_m * p.tiles_n), dim3(64 * WM * WN), lds_bytes, s, p);
    DF_CHECK_LAUNCH();
    return DF_OK;
  }
  if (p.amax_x && p.amax_w && p.w2p) {   // ... with the weights pre-split
    DF_SET_LDS_ONCE((conv_dma_kernel<BM, BN, WM, WN, false, true, true>), (int)lds_bytes);
    hipLaunchKernelGGL((conv_dma_kernel<BM, BN, WM, WN, false, true, true>), dim3(p.tiles_m * p.tiles_n), dim3(64 * WM * WN), lds_bytes, s, p);
    DF_CHECK_LAUNCH();
    return DF_OK;
  }
  if (p.amax_x && p.amax_w) {   // df_conv2d_h2f: fp16x2 on the fragments
    DF_SET_LDS_ONCE((conv_dma_kernel<BM, BN, WM, WN, false, true>), (int)lds_bytes);
    hipLaunchKernelGGL((conv_dma_kernel<BM, BN, WM, WN, false, true>), dim3(p.tiles_m * p.tiles_n), dim3(64 * WM * WN), lds_bytes, s, p);
    DF_CHECK_LAUNCH();
    return DF_OK;
  }
  DF_SET_LDS_ONCE((conv_dma_kernel<BM, BN, WM, WN>), (int)lds_bytes);
  hipLaunchKernelGGL((conv_dma_kernel<BM, BN, WM, WN>), dim3(p.tiles_m * p.tiles_n), dim3(64 * WM * WN), lds_bytes, s, p);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

}  // namespace


// tile variant as BM * 1000 + BN.  rows = GEMM rows one tile range covers (per parity class in class mode).
static int pick_variant(int64_t rows, int64_t rows_per_stat_group, int cout, int epi) {
  if (cout % 64) return 128032;
  static const int force = getenv("DF_CONV_TILE") ? atoi(getenv("DF_CONV_TILE")) : 0;  // A/B experiments only
  if (force == 128064 && rows > 128 * 256 && (!epi_stats(epi) || rows_per_stat_group % 128 == 0)) return 128064;
  int bm;
  if (epi_stats(epi)) bm = (rows_per_stat_group % 128 == 0) ? 128 : 64;
  else {
    // small problems: 64-row tiles, more of them to fill 256 CUs.  Round 6: up to 4096 rows (8192 until then, measured in round 2) -- a B = 1
    // forward's 8192-pixel 3x3 layers are bound by the 64 x 64 tile's operand traffic (split-K over its tap rows changed nothing:
    // profiles/r06_conv_experiments.txt 5m) and run 1.5-2.5 % of the forward faster on the haloed 128-row forms (1.892 / 1.901 -> 1.868 / 1.866 ms)
    static const int small_rows = getenv("DF_CONV_SMALL_ROWS") ? atoi(getenv("DF_CONV_SMALL_ROWS")) : 4096;
    bm = (rows <= small_rows) ? 64 : 128;
  }
  if (bm == 64) return 64064;
  if (cout % 128 == 0) return 128128;
  // 64 output channels: 128x64 (48 KB LDS -> 3 workgroups/CU) measured 120 vs 115 TFLOP/s for the 256x64 tile
  static const int use256 = getenv("DF_CONV_TILE") ? atoi(getenv("DF_CONV_TILE")) == 256064 : 0;
  const bool ok256 = use256 && (epi_stats(epi) ? (rows_per_stat_group % 256 == 0) : (rows >= 256 * 512 && rows % 256 == 0));
  return ok256 ? 256064 : 128064;
}

extern "C" int df_conv2d_variant(int64_t rows, int64_t rows_per_stat_group, int cout, int epi) {
  return pick_variant(rows, rows_per_stat_group, cout, epi);
}

static thread_local int g_last_dma = 0;   // per host thread: the query below refers to the CALLING thread's previous df_conv2d
extern "C" int df_conv2d_last_dma(void) { return g_last_dma; }  // 1 if the previous df_conv2d used the LDS-DMA kernel

extern "C" int df_conv2d_tile_m(int64_t rows_per_stat_group, int cout) {
  return pick_variant(rows_per_stat_group, rows_per_stat_group, cout, DF_EPI_STATS) / 1000;
}

extern "C" int df_conv2d(df_img x, const float* w, const float* bias, df_img y, int ksize, int stride, int pad,
                         int mode, int epi, const float* scale, const float* shift, float* stats_partial,
                         int accumulate, void* stream) {
  return df_conv2d_mp(x, w, bias, y, ksize, stride, pad, mode, epi, scale, shift, stats_partial, accumulate, 0, stream);
}

// w16 != nullptr: the pre-cast bf16 weights of df_conv2d_w16 (w is then unused); query: return 1 / 0 = "the w16 form
// exists for this call" without launching anything
static int conv2d_impl(df_img x, const float* w, const void* w16, const float* bias, df_img y, int ksize, int stride, int pad,
                       int mode, int epi, const float* scale, const float* shift, float* stats_partial,
                       int accumulate, int mfma_bf16, bool query, void* stream, const void* w3 = nullptr,
                       const float* h2_amax_x = nullptr, const float* h2_amax_w = nullptr, float* y_amax = nullptr,
                       const float* y_bound = nullptr, const float* bwd_y = nullptr, const float* bwd_ss = nullptr,
                       const void* w2p = nullptr, const df_img* ups_t = nullptr, int ups_ac = 0);

extern "C" int df_conv2d_mp(df_img x, const float* w, const float* bias, df_img y, int ksize, int stride, int pad,
                            int mode, int epi, const float* scale, const float* shift, float* stats_partial,
                            int accumulate, int mfma_bf16, void* stream) {
  return conv2d_impl(x, w, nullptr, bias, y, ksize, stride, pad, mode, epi, scale, shift, stats_partial, accumulate, mfma_bf16,
                     false, stream);
}

extern "C" int df_conv2d_w16(df_img x, const void* w16, const float* bias, df_img y, int ksize, int stride, int pad,
                             int mode, int epi, const float* scale, const float* shift, float* stats_partial,
                             int accumulate, void* stream) {
  DF_REQUIRE(w16 && df_aligned16(w16), DF_E_ALIGN);
  return conv2d_impl(x, nullptr, w16, bias, y, ksize, stride, pad, mode, epi, scale, shift, stats_partial, accumulate, 1, false,
                     stream);
}

// fp32-accurate convolution through three bf16 planes per operand (conv_halo_x3_kernel): w3 = df_split_bf16x3 of the
// [Cout, 3, 3, Cin] weights (data gradient: of the transposed weights).  Same arguments and epilogues as df_conv2d.
extern "C" int df_conv2d_x3(df_img x, const void* w3, const float* bias, df_img y, int ksize, int stride, int pad,
                            int mode, int epi, const float* scale, const float* shift, float* stats_partial,
                            int accumulate, void* stream) {
  DF_REQUIRE(w3 && df_aligned16(w3), DF_E_ALIGN);
  return conv2d_impl(x, nullptr, nullptr, bias, y, ksize, stride, pad, mode, epi, scale, shift, stats_partial, accumulate, 0, false,
                     stream, w3);
}

extern "C" int df_conv2d_x3_ok(df_img x, df_img y, int ksize, int stride, int mode, int epi) {
  static const int on = getenv("DF_CONV_X3") ? atoi(getenv("DF_CONV_X3")) : 1;
  if (!on || ksize != 3 || stride != 1) return 0;
  const float* any = reinterpret_cast<const float*>(x.ptr);
  const int r = conv2d_impl(x, nullptr, nullptr, nullptr, y, ksize, stride, 1, mode, epi, any, any, const_cast<float*>(any), 0, 0, true,
                            nullptr, x.ptr);
  return r == 1 ? 1 : 0;
}

// fp32-accurate convolution through TWO fp16 planes per operand (conv_halo_x3_kernel<.., NP = 2>): w2 = df_split_h2 of the
// weights with w_amax = df_absmax of them; x_amax = an upper bound of max|x| (df_absmax of x or of any tensor containing it).
// Shapes as df_conv2d_x3 (df_conv2d_x3_ok answers for both).
extern "C" int df_conv2d_h2(df_img x, const void* w2, const float* x_amax, const float* w_amax, const float* bias, df_img y,
                            int ksize, int stride, int pad, int mode, int epi, const float* scale, const float* shift,
                            float* stats_partial, int accumulate, float* y_amax, void* stream) {
  DF_REQUIRE(w2 && df_aligned16(w2) && x_amax && w_amax, DF_E_ALIGN);
  return conv2d_impl(x, nullptr, nullptr, bias, y, ksize, stride, pad, mode, epi, scale, shift, stats_partial, accumulate, 0, false,
                     stream, w2, x_amax, w_amax, y_amax);
}

// ... with PRE-SPLIT tensors (round 4): x may be an h2 image (x.elt = 2, scale = df_h2_scale(*x_amax): x_amax is then the bound
// that DEFINED the planes, not merely an upper bound) and / or y may be written as one (y.elt = 2, scale from *y_bound, a device
// scalar >= max |y| the caller knows before the launch; no accumulate).  df_conv2d_h2p_ok answers whether the pre-split-input
// tile forms exist for a shape (256 x 128 / 512 x 64 tiles: what BASELINE's layers run on).
extern "C" int df_conv2d_h2p(df_img x, const void* w2, const float* x_amax, const float* w_amax, const float* bias, df_img y,
                             const float* y_bound, int ksize, int stride, int pad, int mode, int epi, const float* scale,
                             const float* shift, float* stats_partial, int accumulate, float* y_amax, void* stream) {
  DF_REQUIRE(w2 && df_aligned16(w2) && x_amax && w_amax && epi != DF_EPI_BWD_STATS, DF_E_ALIGN);
  return conv2d_impl(x, nullptr, nullptr, bias, y, ksize, stride, pad, mode, epi, scale, shift, stats_partial, accumulate, 0, false,
                     stream, w2, x_amax, w_amax, y_amax, y_bound);
}

// the 3x3 stride-1 fp16x2 DATA GRADIENT (x = dy: fp32 or pre-split; dz = fp32, contiguous) whose epilogue also leaves the partial
// sums of the BatchNorm + GELU backward of the layer in front -- per tile and channel (sum g, sum g xhat), g = dz gelu'(bn(y)) --
// in bwd_partial ([rows / df_conv2d_tile_m][C][2], the layout df_bn_bwd_finalize reads): bn_y = that layer's conv output (the
// geometry of dz), bn_ss = its statistics per group of dz.grp_size images.  Replaces one full pass over dz and y per layer
// (df_bn_gelu_bwd_reduce).  [REF decoder.py:202-220]: the backward of `nonlinearity(batchnorm(conv(x)))` meeting the next layer's conv
extern "C" int df_conv2d_h2p_dgrad_bn(df_img x, const void* w2, const float* x_amax, const float* w_amax, df_img dz, const float* bn_y,
                                      const float* bn_ss, float* bwd_partial, float* dz_amax, void* stream) {
  DF_REQUIRE(w2 && df_aligned16(w2) && x_amax && w_amax && bn_y && bn_ss && bwd_partial, DF_E_ALIGN);
  return conv2d_impl(x, nullptr, nullptr, nullptr, dz, 3, 1, 1, DF_CONV_DGRAD, DF_EPI_BWD_STATS, nullptr, nullptr, bwd_partial, 0, 0, false,
                     stream, w2, x_amax, w_amax, dz_amax, nullptr, bn_y, bn_ss);
}

extern "C" int df_conv2d_h2p_ok(df_img x, df_img y, int ksize, int stride, int mode, int epi) {
  static const int on = getenv("DF_CONV_X3") ? atoi(getenv("DF_CONV_X3")) : 1;
  if (!on || ksize != 3 || stride != 1) return 0;
  const float* any = reinterpret_cast<const float*>(x.ptr);
  const int r = conv2d_impl(x, nullptr, nullptr, nullptr, y, ksize, stride, 1, mode, epi, any, any, const_cast<float*>(any), 0, 0, true,
                            nullptr, x.ptr, any, any, nullptr, any);
  return r == 1 ? 1 : 0;
}

// the fp32-input kernels of df_conv2d_mp / df_conv2d_h2f / df_conv2d_amax with an h2 OUTPUT (y.elt = 2, scale from *y_bound): the
// 1x1 convolutions that write a half of a pre-split concatenation, the 1x1 data gradients that feed a pre-split 3x3 layer
// df_conv2d_h2f / df_conv2d_yh2 (y.elt = 2: y_bound required, no y_amax) with the weights ALSO given pre-split: w2 = the [hi | lo]
// fp16 planes of w scaled by df_h2_scale(*w_amax) (df_split_h2, or the layer's planes from df_weight_prep) -- conv_dma_kernel<.., H2,
// BP> fetches them by DMA instead of splitting the fp32 weight fragments in every wave; w (fp32) is still what the tile forms
// without an fp16x2 mode read.
extern "C" int df_conv2d_h2f_wp(df_img x, const float* w, const void* w2, const float* x_amax, const float* w_amax, const float* bias,
                                df_img y, const float* y_bound, int ksize, int stride, int pad, int mode, int epi, const float* scale,
                                const float* shift, float* stats_partial, int accumulate, float* y_amax, void* stream) {
  DF_REQUIRE(x_amax && w_amax && w2 && df_aligned16(w2) && (y.elt != 2 || y_bound), DF_E_ARG);
  return conv2d_impl(x, w, nullptr, bias, y, ksize, stride, pad, mode, epi, scale, shift, stats_partial, accumulate, 0, false, stream,
                     nullptr, x_amax, w_amax, y.elt == 2 ? nullptr : y_amax, y.elt == 2 ? y_bound : nullptr, nullptr, nullptr, w2);
}

// df_conv2d_h2f_wp for the 1x1 skip convolution of an UpsampleSkip block (forward, bias epilogue, y = the SECOND half of the pre-split
// concatenation: channels [t.c, 2 t.c) of a pixel whose first t.c channels precede y.ptr) that ALSO writes the first half: the bilinear
// x2 of t [N, H / 2, W / 2, C] (fp32; align_corners as F.interpolate), scaled by the same bound *y_bound (>= max |t| as well) -- one
// kernel writes whole pixels of the concatenation instead of two kernels one half each.  Replaces df_upsample2x_h2 + df_conv2d_h2f_wp;
// DF_E_SHAPE where the 8-wave DMA kernels with pre-split weights do not cover the call (the caller then issues the two launches).
extern "C" int df_conv2d_h2f_wp_up(df_img x, const float* w, const void* w2, const float* x_amax, const float* w_amax, const float* bias,
                                   df_img y, const float* y_bound, df_img t, int align_corners, void* stream) {
  DF_REQUIRE(x_amax && w_amax && w2 && df_aligned16(w2) && y.elt == 2 && y_bound, DF_E_ARG);
  return conv2d_impl(x, w, nullptr, bias, y, 1, 1, 0, DF_CONV_FWD, DF_EPI_BIAS, nullptr, nullptr, nullptr, 0, 0, false, stream, nullptr, x_amax,
                     w_amax, nullptr, y_bound, nullptr, nullptr, w2, &t, align_corners);
}

extern "C" int df_conv2d_yh2(df_img x, const float* w, const float* x_amax, const float* w_amax, const float* bias, df_img y,
                             const float* y_bound, int ksize, int stride, int pad, int mode, int epi, const float* scale,
                             const float* shift, float* stats_partial, int accumulate, void* stream) {
  DF_REQUIRE(y.elt == 2 && y_bound, DF_E_ARG);
  return conv2d_impl(x, w, nullptr, bias, y, ksize, stride, pad, mode, epi, scale, shift, stats_partial, accumulate, 0, false, stream,
                     nullptr, x_amax, w_amax, nullptr, y_bound);
}

// fp16x2 for the convolutions WITHOUT a haloed form (1x1, stride 2; conv_dma_kernel<.., H2>): fp32 weights as they are (the
// fragments are split in registers), x_amax / w_amax as for df_conv2d_h2.  Shapes the 8-wave DMA kernels do not cover run the fp32
// kernels of df_conv2d (same result class, no error).
extern "C" int df_conv2d_h2f(df_img x, const float* w, const float* x_amax, const float* w_amax, const float* bias, df_img y,
                             int ksize, int stride, int pad, int mode, int epi, const float* scale, const float* shift,
                             float* stats_partial, int accumulate, float* y_amax, void* stream) {
  DF_REQUIRE(x_amax && w_amax, DF_E_ARG);
  return conv2d_impl(x, w, nullptr, bias, y, ksize, stride, pad, mode, epi, scale, shift, stats_partial, accumulate, 0, false, stream,
                     nullptr, x_amax, w_amax, y_amax);
}

// df_conv2d (fp32 MFMA kernels, any supported shape) that also leaves max |y| in *y_amax (zero-initialised by the caller; see
// df_absmax): the 1x1 and stride-2 convolutions whose output an fp16x2 convolution reads next
extern "C" int df_conv2d_amax(df_img x, const float* w, const float* bias, df_img y, int ksize, int stride, int pad, int mode, int epi,
                              const float* scale, const float* shift, float* stats_partial, int accumulate, float* y_amax,
                              void* stream) {
  return conv2d_impl(x, w, nullptr, bias, y, ksize, stride, pad, mode, epi, scale, shift, stats_partial, accumulate, 0, false, stream,
                     nullptr, nullptr, nullptr, y_amax);
}

extern "C" int df_absmax(df_img x, float* amax, void* stream) {
  DF_REQUIRE(img_ok(x) && amax && (x.c % 4) == 0, DF_E_ARG);
  const int64_t total4 = (int64_t)x.n * x.h * x.w * (x.c / 4);
  int64_t blocks = (total4 + 1023) / 1024;          // ~4 float4 per thread
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, total4,
                     reinterpret_cast<unsigned*>(amax));
  DF_CHECK_LAUNCH();
  return DF_OK;
}

extern "C" int df_split_h2(const float* w, const float* amax, void* out2, int64_t n, void* stream) {
  DF_REQUIRE(w && amax && out2 && n > 0 && df_aligned16(out2), DF_E_ARG);
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(split_h2_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w, amax,
                     reinterpret_cast<_Float16*>(out2), n);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

extern "C" int df_split_bf16x3(const float* w, void* out3, int64_t n, void* stream) {
  DF_REQUIRE(w && out3 && n > 0 && df_aligned16(out3), DF_E_ARG);
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(split_bf16x3_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w,
                     reinterpret_cast<__bf16*>(out3), n);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

extern "C" int df_conv2d_w16_ok(df_img x, df_img y, int ksize, int stride, int mode, int epi) {
  static const int on = getenv("DF_CONV_W16") ? atoi(getenv("DF_CONV_W16")) : 1;
  if (!on || (ksize != 1 && ksize != 3)) return 0;
  const float* any = reinterpret_cast<const float*>(x.ptr);   // argument checks only: nothing is dereferenced in query mode
  const int r = conv2d_impl(x, nullptr, x.ptr, nullptr, y, ksize, stride, ksize / 2, mode, epi, any, any, const_cast<float*>(any), 0,
                            1, true, nullptr);
  return r == 1 ? 1 : 0;
}

static int conv2d_impl(df_img x, const float* w, const void* w16, const float* bias, df_img y, int ksize, int stride, int pad,
                       int mode, int epi, const float* scale, const float* shift, float* stats_partial,
                       int accumulate, int mfma_bf16, bool query, void* stream, const void* w3, const float* h2_amax_x,
                       const float* h2_amax_w, float* y_amax, const float* y_bound, const float* bwd_y, const float* bwd_ss,
                       const void* w2p, const df_img* ups_t, int ups_ac) {
  if (w3) w = reinterpret_cast<const float*>(w3);   // (argument checks below want a non-null, aligned weight pointer)
  // bfloat16 tensors (bf16-storage training): the input only for the bf16-tile kernel (df_conv2d_w16), the output for any
  // kernel with the branch-free epilogue
  DF_REQUIRE(img_ok(x, w16 != nullptr || (w3 && h2_amax_x)) && img_ok(y, true) && (w16 || (w && df_aligned16(w))), DF_E_ALIGN);
  DF_REQUIRE((x.elt != 2 || (w3 && h2_amax_x)) && (x.elt != 1 || w16), DF_E_ARG);      // h2 input: the fp16x2 haloed form only
  // h2 output: a bound, no accumulation (the planes of two scales do not add), whole 32-channel chunks at 128-byte lines
  if (y.elt == 2)
    DF_REQUIRE((query || y_bound) && !accumulate && (y.ld % 32) == 0 && (y.img_stride % 32) == 0 && (y.grp_off % 32) == 0 &&
               (((uintptr_t)y.ptr) & 127) == 0, DF_E_ARG);
  if (x.elt == 2)
    DF_REQUIRE((x.ld % 32) == 0 && (x.img_stride % 32) == 0 && (x.grp_off % 32) == 0 && (((uintptr_t)x.ptr) & 127) == 0, DF_E_ARG);
  const int xes = x.elt == 1 ? 2 : 4, yes = y.elt == 1 ? 2 : 4;
  DF_REQUIRE(x.n == y.n, DF_E_SHAPE);
  DF_REQUIRE((ksize == 1 || ksize == 3) && (stride == 1 || stride == 2) && pad == ksize / 2, DF_E_SHAPE);
  DF_REQUIRE(mode == DF_CONV_FWD || mode == DF_CONV_DGRAD, DF_E_ARG);
  DF_REQUIRE(epi >= 0 && epi <= 3, DF_E_ARG);
  DF_REQUIRE(x.c % BK == 0 && y.c % 32 == 0, DF_E_SHAPE);
  if (mode == DF_CONV_FWD) {
    DF_REQUIRE(y.h == (x.h + 2 * pad - ksize) / stride + 1 && y.w == (x.w + 2 * pad - ksize) / stride + 1, DF_E_SHAPE);
  } else {
    DF_REQUIRE(x.h == (y.h + 2 * pad - ksize) / stride + 1 && x.w == (y.w + 2 * pad - ksize) / stride + 1, DF_E_SHAPE);
  }
  if (epi == DF_EPI_BN_GELU) DF_REQUIRE(scale && shift, DF_E_ARG);
  if (epi_stats(epi)) DF_REQUIRE(stats_partial, DF_E_ARG);
  if (epi == DF_EPI_BWD_STATS)      // fused BatchNorm-backward partials: the fp16x2 data gradient with a plain fp32 output only
    DF_REQUIRE(mode == DF_CONV_DGRAD && w3 && h2_amax_x && x.elt == 2 && (query || (bwd_y && bwd_ss)) && y.elt == 0 && !accumulate && y.ld == y.c &&
                   !bias, DF_E_ARG);
  ConvParams p;
  p.x = x; p.y = y; p.w = w; p.bias = bias; p.scale = scale; p.shift = shift; p.stats = stats_partial;
  p.ks = ksize; p.stride = stride; p.pad = pad; p.mode = mode; p.epi = epi; p.accumulate = accumulate;
  p.bf16 = mfma_bf16 != 0;
  p.stats_mul = 1;
  p.amax_x = h2_amax_x;      // (used by the fp16x2 forms only: conv_halo_x3_kernel<NP = 2> below, conv_dma_kernel<.., H2>)
  p.amax_w = h2_amax_w;
  p.w2p = w2p;
  p.amax_y = reinterpret_cast<unsigned*>(y_amax);
  p.bound_y = y_bound;
  p.bwd_y = bwd_y;
  p.bwd_ss = bwd_ss;
  static const int conv_rot = getenv("DF_CONV_ROT") ? atoi(getenv("DF_CONV_ROT")) : 1;
  p.rot = conv_rot;
  static const int cls_il = getenv("DF_CONV_CLS_IL") ? atoi(getenv("DF_CONV_CLS_IL")) : 1;
  p.cls_il = cls_il;
  p.ups_on = 0; p.ups_ac = ups_ac;
  if (ups_t) {   // df_conv2d_h2f_wp_up: a 1x1 forward conv into the second half of a pre-split concatenation, fp32 t of half the resolution
    const df_img& t = *ups_t;
    DF_REQUIRE(!w3 && !w16 && ksize == 1 && stride == 1 && mode == DF_CONV_FWD && epi == DF_EPI_BIAS && !accumulate && y.elt == 2 && x.elt == 0 &&
                   w2p && h2_amax_x && h2_amax_w && !mfma_bf16, DF_E_ARG);
    DF_REQUIRE(t.ptr && df_aligned16(t.ptr) && t.elt == 0 && t.n == y.n && t.c == y.c && 2 * t.h == y.h && 2 * t.w == y.w && (t.c % 8) == 0 &&
                   (t.ld % 4) == 0 && (t.img_stride % 4) == 0 && (t.grp_off % 4) == 0 && t.grp_size > 0 && (y.ld % 32) == 0 && y.ld >= 2 * y.c,
               DF_E_SHAPE);
    p.ups_t = t;
    p.ups_on = 1;
  }
  p.hw_y = y.h * y.w;
  const int64_t M = (int64_t)y.n * p.hw_y;
  DF_REQUIRE(M < (1ll << 31), DF_E_SHAPE);
  p.M = (int)M; p.K = x.c; p.N = y.c;
  p.cls_tiles = 0;
  static const int dbg_env = getenv("DF_CONV_DBG") ? atoi(getenv("DF_CONV_DBG")) : 0;
  p.dbg = dbg_env;
  const int64_t rows_per_group = (int64_t)y.grp_size * p.hw_y;
  int64_t rows = M;
  const bool cls = mode == DF_CONV_DGRAD && stride == 2 && ksize == 3 && epi == DF_EPI_BIAS && (y.h % 2) == 0 && (y.w % 2) == 0;
  if (cls) rows = M / 4;
  int var = pick_variant(rows, rows_per_group, p.N, epi);
  int bm = var / 1000;
  if (epi_stats(epi)) DF_REQUIRE(rows_per_group % bm == 0, DF_E_SHAPE);
  if (cls && rows % bm == 0) {
    p.cls_tiles = (int)(rows / bm);
    p.tiles_m = 4 * p.cls_tiles;
  } else {
    if (cls) { var = pick_variant(M, rows_per_group, p.N, epi); bm = var / 1000; }
    p.tiles_m = (int)((M + bm - 1) / bm);
  }
  p.tiles_n = p.N / (var % 1000);
  // LDS-DMA path needs 32-bit byte offsets: extent of x (+ the shift that keeps the top-left tap offset >= 0)
  p.x_bytes = p.w_bytes = p.dshift = 0;
  {
    static const int no_dma = getenv("DF_CONV_NO_DMA") ? atoi(getenv("DF_CONV_NO_DMA")) : 0;
    const int64_t groups = x.n / x.grp_size;
    const int64_t ext = ((int64_t)(x.grp_size - 1) * x.img_stride + (groups - 1) * x.grp_off + (int64_t)x.h * x.w * x.ld) * xes;
    const int64_t dsh = (p.cls_tiles > 0) ? 0 : (int64_t)pad * ((int64_t)x.w + 1) * x.ld * xes;
    const int64_t wb = (int64_t)p.N * ksize * ksize * p.K * 4;
    const bool geom_ok = !(mode == DF_CONV_DGRAD && stride == 2 && p.cls_tiles == 0);  // generic s2 dgrad: register path
    if (!no_dma && geom_ok && x.img_stride >= 0 && x.grp_off >= 0 && ext + dsh < (int64_t)DMA_BAD - (16 << 20) && wb < (1ll << 31) &&
        ((int64_t)2 * x.w + 2) * x.ld * 4 + (int64_t)p.K * 4 < (8 << 20)) {
      p.x_bytes = (unsigned)(ext + dsh);
      p.w_bytes = (unsigned)wb;
      p.dshift = (unsigned)dsh;
    }
  }
  {
    static const int wide_epi = getenv("DF_CONV_WIDE_EPI") ? atoi(getenv("DF_CONV_WIDE_EPI")) : 0;
    const int64_t ygroups = y.n / y.grp_size;
    const int64_t yext = ((int64_t)(y.grp_size - 1) * y.img_stride + (ygroups - 1) * y.grp_off + (int64_t)y.h * y.w * y.ld) * yes;
    p.y_bytes = ((!wide_epi || y.elt) && y.img_stride >= 0 && y.grp_off >= 0 && yext < (int64_t)0xFFFFFFFFll - (16 << 20)) ? (unsigned)yext : 0u;
    if (y.elt) DF_REQUIRE(p.y_bytes != 0 && p.x_bytes != 0, DF_E_SHAPE);   // bf16 output: only the DMA kernels' straight-line epilogue
  }
  if (!query) g_last_dma = p.x_bytes != 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // haloed-A kernel: 3x3 stride 1 (fwd / dgrad), rows of 128 output pixels inside one image row, full tiles only
  static const int use_halo = getenv("DF_CONV_HALO") ? atoi(getenv("DF_CONV_HALO")) : 1;
  const bool halo_ok = use_halo && p.x_bytes && ksize == 3 && stride == 1 && p.cls_tiles == 0 && (y.w % 128) == 0 &&
                       x.w == y.w && x.h == y.h && (var == 128128 || var == 128064);
  if (w3) {    // df_conv2d_x3: fp32-accurate product from three bf16 planes per operand (W % 128 == 0, or W == 64 as row pairs)
    const bool two = p.x_bytes && ksize == 3 && stride == 1 && p.cls_tiles == 0 && y.w == 64 && (y.h % 2) == 0 && x.w == y.w &&
                     x.h == y.h && (var == 128128 || var == 128064);
    const bool h2 = h2_amax_x != nullptr;        // two fp16 planes (df_conv2d_h2) instead of three bf16 ones
    const bool xp = x.elt == 2;                  // pre-split input (h2 image): the 256 x 128 / 512 x 64 tile forms below only
    bool ok = (halo_ok || two) && (p.K % BK) == 0 && (x.elt == 0 || (xp && h2)) && (y.elt == 0 || (y.elt == 2 && h2));
    if (ok && xp) {
      const int seg_ = (y.w % 256) == 0 ? 1 : y.w == 128 ? 2 : y.w == 64 ? 4 : 0;
      const int seg64_ = (y.w % 512) == 0 ? 1 : y.w == 256 ? 2 : 0;
      const bool big128 = var == 128128 && seg_ && (y.h % seg_) == 0 && (M % 256) == 0 && (!epi_stats(epi) || rows_per_group % 256 == 0) &&
                          M / 256 * p.tiles_n >= 512;
      const bool big64 = var != 128128 && !two && seg64_ && (y.h % seg64_) == 0 && (M % 512) == 0 &&
                         (!epi_stats(epi) || rows_per_group % 512 == 0) && M / 512 >= 512;
      ok = big128 || big64;
    }
    if (query) return ok ? 1 : 0;
    DF_REQUIRE(ok, DF_E_SHAPE);
    p.w = reinterpret_cast<const float*>(w3);
    p.w_bytes = (unsigned)((int64_t)p.N * 9 * p.K * 2 * (h2 ? 2 : 3));
    p.bf16 = 0;
    p.amax_x = h2_amax_x;
    p.amax_w = h2_amax_w;
    static const int bm256 = getenv("DF_CONV_X3_BM256") ? atoi(getenv("DF_CONV_X3_BM256")) : 1;
    // 64 output channels: 256-pixel row tiles where the image allows (wave tile 64 x 32 instead of 32 x 32)
    const bool wide = !two && var != 128128 && bm256 && (y.w % 256) == 0 && (M % 256) == 0 && (!epi_stats(epi) || rows_per_group % 256 == 0);
    if (wide) {
      p.tiles_m = (int)(M / 256);
      p.stats_mul = 2;
    }
    if (h2) {
      // 128-multiple output channels: 256 x 128 tiles whose waves own 64 x 64 (8 fragment reads per 12 MFMAs instead of 6 per 6 --
      // the 128 x 128 form reads LDS for as many cycles as it multiplies) where the rows allow: 256 pixels of one image row, two
      // rows of a W == 128 image, four of a W == 64 one
      static const int big = getenv("DF_CONV_H2_BM256") ? atoi(getenv("DF_CONV_H2_BM256")) : 1;
      const int seg = (y.w % 256) == 0 ? 1 : y.w == 128 ? 2 : y.w == 64 ? 4 : 0;
      if (big && var == 128128 && seg && (y.h % seg) == 0 && (M % 256) == 0 && (!epi_stats(epi) || rows_per_group % 256 == 0) &&
          M / 256 * p.tiles_n >= 512) {
        p.tiles_m = (int)(M / 256);
        p.stats_mul = 2;
        static const int pers = getenv("DF_CONV_PERS") ? atoi(getenv("DF_CONV_PERS")) : 1;
        if (xp && pers && epi == DF_EPI_BWD_STATS) {
          return df_launch_conv_halo_x3p(p, 128, seg, true, s);
        }
        if (xp && pers) {
          return df_launch_conv_halo_x3p(p, 128, seg, false, s);
        }
        if (xp && epi == DF_EPI_BWD_STATS) {
          if (seg == 1) return launch_conv_halo_x3<256, 128, 4, 2, 1, 4, 2, false, true, true>(p, s);
          if (seg == 2) return launch_conv_halo_x3<256, 128, 4, 2, 2, 4, 2, false, true, true>(p, s);
          return launch_conv_halo_x3<256, 128, 4, 2, 4, 4, 2, false, true, true>(p, s);
        }
        if (xp) {
          if (seg == 1) return launch_conv_halo_x3<256, 128, 4, 2, 1, 4, 2, false, true>(p, s);
          if (seg == 2) return launch_conv_halo_x3<256, 128, 4, 2, 2, 4, 2, false, true>(p, s);
          return launch_conv_halo_x3<256, 128, 4, 2, 4, 4, 2, false, true>(p, s);
        }
        if (seg == 1) return launch_conv_halo_x3<256, 128, 4, 2, 1, 4, 2>(p, s);
        if (seg == 2) return launch_conv_halo_x3<256, 128, 4, 2, 2, 4, 2>(p, s);
        return launch_conv_halo_x3<256, 128, 4, 2, 4, 4, 2>(p, s);
      }
      // 64 output channels: 512 x 64 tiles (8 x 1 waves of 64 x 64): 512 pixels of one row, or two rows of a W == 256 image
      const int seg64 = (y.w % 512) == 0 ? 1 : y.w == 256 ? 2 : 0;
      if (big && var != 128128 && !two && seg64 && (y.h % seg64) == 0 && (M % 512) == 0 && (!epi_stats(epi) || rows_per_group % 512 == 0) &&
          M / 512 >= 512) {
        p.tiles_m = (int)(M / 512);
        p.stats_mul = 4;
        static const int pers64 = getenv("DF_CONV_PERS") ? atoi(getenv("DF_CONV_PERS")) : 1;
        if (xp && pers64 && epi == DF_EPI_BWD_STATS)
          return df_launch_conv_halo_x3p(p, 64, seg64, true, s);
        if (xp && pers64) return df_launch_conv_halo_x3p(p, 64, seg64, false, s);
        if (xp && epi == DF_EPI_BWD_STATS)
          return seg64 == 1 ? launch_conv_halo_x3<512, 64, 8, 1, 1, 3, 2, false, true, true>(p, s) : launch_conv_halo_x3<512, 64, 8, 1, 2, 3, 2, false, true, true>(p, s);
        if (xp) return seg64 == 1 ? launch_conv_halo_x3<512, 64, 8, 1, 1, 3, 2, false, true>(p, s) : launch_conv_halo_x3<512, 64, 8, 1, 2, 3, 2, false, true>(p, s);
        return seg64 == 1 ? launch_conv_halo_x3<512, 64, 8, 1, 1, 3, 2>(p, s) : launch_conv_halo_x3<512, 64, 8, 1, 2, 3, 2>(p, s);
      }
      DF_REQUIRE(!xp, DF_E_SHAPE);
      if (two) return var == 128128 ? launch_conv_halo_x3<128, 128, 2, 4, 2, 4, 2>(p, s) : launch_conv_halo_x3<128, 64, 4, 2, 2, 8, 2>(p, s);
      if (var == 128128) return launch_conv_halo_x3<128, 128, 2, 4, 1, 4, 2>(p, s);
      if (wide) return launch_conv_halo_x3<256, 64, 4, 2, 1, 4, 2>(p, s);
      return launch_conv_halo_x3<128, 64, 4, 2, 1, 8, 2>(p, s);
    }
    if (two) return var == 128128 ? launch_conv_halo_x3<128, 128, 2, 4, 2, 4>(p, s) : launch_conv_halo_x3<128, 64, 4, 2, 2, 8>(p, s);
    if (var == 128128) return launch_conv_halo_x3<128, 128, 2, 4, 1, 4>(p, s);
    if (wide) return launch_conv_halo_x3<256, 64, 4, 2, 1, 4>(p, s);
    return launch_conv_halo_x3<128, 64, 4, 2, 1, 8>(p, s);
  }
  if (w16) {   // df_conv2d_w16: bf16 tiles in LDS, only the haloed forms exist (W % 128 == 0, or W == 64 as row pairs)
    const bool two = p.x_bytes && ksize == 3 && stride == 1 && p.cls_tiles == 0 && y.w == 64 && (y.h % 2) == 0 && x.w == y.w &&
                     x.h == y.h && (var == 128128 || var == 128064);
    const bool ok = (halo_ok || two) && (p.K % BK) == 0;
    if (query) return ok ? 1 : 0;
    DF_REQUIRE(ok, DF_E_SHAPE);
    p.w = reinterpret_cast<const float*>(w16);
    p.w_bytes = p.w_bytes / 2;
    p.bf16 = 1;
    static const int w4 = getenv("DF_W16_WAVES4") ? atoi(getenv("DF_W16_WAVES4")) : 0;   // bit 0: 128-wide, bit 1: 64-wide tiles on 4 waves
    {   // DF_W16_WIDE=1 (A/B; default off): the large tiles of the fp16x2 kernel with ONE bf16 plane (64 x 64 wave tiles: 4 fragment
        // reads per 4 MFMAs; the 128 x 128 forms below read 3 per 2).  Measured SLOWER than the forms below at the bench shape
        // (bf16 step 42.8 vs 41.4 ms; 700-930 vs 820-1000 TFLOP/s per layer): with one plane the kernel is not LDS-read-bound, and
        // this kernel's register-staged halo costs more than conv_halo_w16_kernel's
      static const int wide = getenv("DF_W16_WIDE") ? atoi(getenv("DF_W16_WIDE")) : 0;
      const int seg = (y.w % 256) == 0 ? 1 : y.w == 128 ? 2 : y.w == 64 ? 4 : 0;
      if (wide && var == 128128 && seg && (y.h % seg) == 0 && (M % 256) == 0 && (!epi_stats(epi) || rows_per_group % 256 == 0) &&
          M / 256 * p.tiles_n >= 512) {
        p.tiles_m = (int)(M / 256);
        p.stats_mul = 2;
        if (x.elt) {
          if (seg == 1) return launch_conv_halo_x3<256, 128, 4, 2, 1, 8, 1, true>(p, s);
          if (seg == 2) return launch_conv_halo_x3<256, 128, 4, 2, 2, 8, 1, true>(p, s);
          return launch_conv_halo_x3<256, 128, 4, 2, 4, 8, 1, true>(p, s);
        }
        if (seg == 1) return launch_conv_halo_x3<256, 128, 4, 2, 1, 8, 1, false>(p, s);
        if (seg == 2) return launch_conv_halo_x3<256, 128, 4, 2, 2, 8, 1, false>(p, s);
        return launch_conv_halo_x3<256, 128, 4, 2, 4, 8, 1, false>(p, s);
      }
      const int seg64 = (y.w % 512) == 0 ? 1 : y.w == 256 ? 2 : 0;    // 64 output channels: 512 x 64 tiles, 8 x 1 waves
      if (wide && var == 128064 && seg64 && (y.h % seg64) == 0 && (M % 512) == 0 && (!epi_stats(epi) || rows_per_group % 512 == 0) &&
          M / 512 >= 512) {
        p.tiles_m = (int)(M / 512);
        p.stats_mul = 4;
        if (x.elt) return seg64 == 1 ? launch_conv_halo_x3<512, 64, 8, 1, 1, 8, 1, true>(p, s) : launch_conv_halo_x3<512, 64, 8, 1, 2, 8, 1, true>(p, s);
        return seg64 == 1 ? launch_conv_halo_x3<512, 64, 8, 1, 1, 8, 1, false>(p, s) : launch_conv_halo_x3<512, 64, 8, 1, 2, 8, 1, false>(p, s);
      }
    }
    if (x.elt) {   // bf16 activations in memory: 8-wave forms only
      if (halo_ok) return var == 128128 ? launch_conv_halo_w16<128, 2, 4, 1, true>(p, s) : launch_conv_halo_w16<64, 4, 2, 1, true>(p, s);
      return var == 128128 ? launch_conv_halo_w16<128, 2, 4, 2, true>(p, s) : launch_conv_halo_w16<64, 4, 2, 2, true>(p, s);
    }
    if (halo_ok) {
      if (var == 128128) return (w4 & 1) ? launch_conv_halo_w16<128, 2, 2>(p, s) : launch_conv_halo_w16<128, 2, 4>(p, s);
      return (w4 & 2) ? launch_conv_halo_w16<64, 2, 2>(p, s) : launch_conv_halo_w16<64, 4, 2>(p, s);
    }
    if (var == 128128) return (w4 & 1) ? launch_conv_halo_w16<128, 2, 2, 2>(p, s) : launch_conv_halo_w16<128, 2, 4, 2>(p, s);
    return (w4 & 2) ? launch_conv_halo_w16<64, 2, 2, 2>(p, s) : launch_conv_halo_w16<64, 4, 2, 2>(p, s);
  }
  DF_REQUIRE(x.elt == 0, DF_E_ARG);
  if (p.ups_on) {      // only the 8-wave DMA kernels with pre-split weights carry the UPS form
    DF_REQUIRE(p.x_bytes && p.w2p && p.amax_x && p.amax_w && !p.bf16, DF_E_SHAPE);
    switch (var) {
      case 128032: return launch_conv_w8<128, 32, 4, 1>(p, s);
      case 64064: return launch_conv_w8<64, 64, 2, 2>(p, s);
      case 128128: return launch_conv_w8<128, 128, 4, 2>(p, s);
      case 256064: return launch_conv_w8<256, 64, 4, 2>(p, s);
      default: return launch_conv_w8<128, 64, 4, 2>(p, s);
    }
  }
  switch (var) {
    // (round 5: the small tiles chosen for layers with few pixels -- a B = 1 forward -- ran on the fp32 MFMA whatever the caller asked
    // for: launch_conv has no fp16x2 form.  With a bound on x and w they take the DMA-tile kernel's H2 forms like the large tiles.)
    case 128032:
      if (p.x_bytes && p.amax_x && p.amax_w) return launch_conv_w8<128, 32, 4, 1>(p, s);
      return launch_conv<128, 32, 4, 1>(p, s);
    case 64064:
      if (p.x_bytes && p.amax_x && p.amax_w) return launch_conv_w8<64, 64, 2, 2>(p, s);
      return launch_conv<64, 64, 2, 2>(p, s);
    case 128128: {
      static const int w8 = getenv("DF_CONV_W8") ? atoi(getenv("DF_CONV_W8")) : 3;
      if (halo_ok) return launch_conv_halo<128, 2, 4>(p, s);
      // pre-split weights (BP): B fragments cost no VALU, so the wave grid that shares an A fragment among FEWER waves wins -- 4 x 2
      // (each wave splits one 32-row A fragment per 6 MFMAs) instead of 2 x 4 (two per 6, each redone by four waves)
      static const int bp42 = getenv("DF_CONV_BP42") ? atoi(getenv("DF_CONV_BP42")) : 1;
      if ((w8 & 1) && p.x_bytes && p.w2p && p.amax_x && p.amax_w && bp42) return launch_conv_w8<128, 128, 4, 2>(p, s);
      if ((w8 & 1) && p.x_bytes) return launch_conv_w8<128, 128, 2, 4>(p, s);
      return launch_conv<128, 128, 2, 2>(p, s);
    }
    case 256064: {
      static const int w8 = getenv("DF_CONV_W8") ? atoi(getenv("DF_CONV_W8")) : 3;
      if ((w8 & 4) && p.x_bytes) return launch_conv_w8<256, 64, 4, 2>(p, s);
      return launch_conv<256, 64, 4, 1>(p, s);
    }
    default: {
      static const int w8 = getenv("DF_CONV_W8") ? atoi(getenv("DF_CONV_W8")) : 3;
      if (halo_ok) return launch_conv_halo<64, 4, 2>(p, s);
      // (with pre-split weights a 4 x 1 grid -- 32 x 64 wave tiles, one A split per 6 MFMAs, but 4 waves per workgroup -- measured
      // 9-33 % SLOWER than 4 x 2 on the four 64-wide layers: occupancy beats the split count here)
      if ((w8 & 2) && p.x_bytes) return launch_conv_w8<128, 64, 4, 2>(p, s);
      return launch_conv<128, 64, 2, 2>(p, s);
    }
  }
}


