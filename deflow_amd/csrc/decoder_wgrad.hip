// Fused weight gradients of the ConvGRU gates ([REF decoder.py:123-147] differentiated w.r.t. convz/convr/convq).
//
//   dW_z = dz_pre^T [h_in | x]     dW_r = dr_pre^T [h_in | x]     dW_q = dq_pre^T [r*h | x]      (each [128, 192])
//
// summed over every valid point row of every GRU iteration -- three "TN" GEMMs with a 4.6 M-deep reduction at the
// bench shape.  As six generic 1x1 weight-gradient calls these were HBM-latency-bound streams (two operand planes
// per 128 x 128 tile, 32 flops per byte, one chunk of prefetch): 12 ms per step, as much as the data-gradient kernel.
// Here one launch streams each plane once per tile: a workgroup owns one gate's full [128 co x 192 ci] tile (49 k
// flops per 1280 operand bytes) over a contiguous range of 16-row chunks, operands go global -> LDS by DMA
// (buffer_load ... lds) through a 3-deep ring, i.e. two stages (~5 us) of prefetch with no staging registers, and
// only chunks holding valid rows are visited.  Partial tiles go to ws[split][384][192]; df_conv2d_wgrad_reduce sums
// them in a fixed order (deterministic).
#include "common.h"

namespace {

constexpr int WP = 16;                          // rows per stage
constexpr int WD_F32 = 3, STG_F32 = WP * (128 + 128 + 64);   // fp32 planes: ring depth, floats per stage G[16][128] H[16][128] X[16][64]
// bf16 planes: G and H tiles are half the size.  Measured at the bench shape (bf16 mode, 2.40 ms before): bf16 planes alone 2.40
// (not HBM-bound), the wave-uniform H / X branch hoisted 1.98 (a per-element select between the bf16 and the fp32 tile had
// compiled to a scalar branch per LDS read), gates of a split on one XCD 1.99, a 5-deep ring in the same 60 KB 2.01 (not
// latency-bound either): what is left is the 40 scalar LDS reads per wave and stage behind 6 MFMAs.
constexpr int WD_BF = 3, STG_BF = WP * (64 + 64 + 64);
constexpr unsigned BAD = 0xFFFFFFFFu - (8u << 20);  // always outside the buffer range -> the DMA writes zeros
typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct GruWgradParams {
  const float* save;   // planes as written by the forward / backward kernels: [6][T][B*N][128]; bf16 mode: bf16 half rows
  const float* x;      // [B*N][64] offset encoding
  const int32_t* counts;
  int B, N, T, nsplit;
  int64_t plane_stride, iter_stride;  // floats
  float* ws;           // [nsplit][384][192]
};

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

// X2 (fp32 training, mfma_bf16 == 3; round 3): fp32 planes and tiles as in the fp32 form, but every operand fragment is split in
// registers into two bf16 planes (hi + lo, 16 significant bits) and each product takes three v_mfma_f32_32x32x16_bf16 (lo hi',
// hi lo', hi hi') instead of eight v_mfma_f32_32x32x2_f32: 18 MFMAs of 32 cycles per 16-row stage instead of 48 of 64.  The
// 2^-16 relative error per product is random and the sums run over millions of rows.
template <bool BF, bool X2 = false>
__global__ __launch_bounds__(256, 2) void gru_wgrad_kernel(GruWgradParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int WD = BF ? WD_BF : WD_F32, STG = BF ? STG_BF : STG_F32, OPS = BF ? 3 : 5;   // OPS: DMA instructions per wave and stage
  constexpr int GSZ = BF ? WP * 64 : WP * 128;   // floats of the G (and H) tile
  __shared__ __attribute__((aligned(16))) float ring[WD * STG];   // 60 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kh = lane >> 5;
  const int wco = wave & 1, wci = wave >> 1;
  // The three gates of a split stream the same x rows (z and r also the same h_in rows): XCD-aware order -- every XCD gets a
  // contiguous range of (split, gate) pairs, the gates of a split back to back -- lets two of the three reads hit that L2.
  const int lg = df_xcd_swizzle(blockIdx.x, gridDim.x);
  const int split = lg / 3, gate = lg - 3 * split;   // gate 0: z, 1: r, 2: q
  const float* gplane = p.save + (1 + gate) * p.plane_stride;
  const float* hplane = p.save + (gate == 2 ? 4 : 0) * p.plane_stride;

  // ---- this workgroup's range of valid 16-row chunks, enumerated (iteration, sample, chunk) -----------------------
  int S = 0;
  for (int b = 0; b < p.B; ++b) S += (p.counts[b] + WP - 1) / WP;
  const int64_t total = (int64_t)S * p.T;
  const int64_t w0 = total * split / p.nsplit, w1 = total * (split + 1) / p.nsplit;
  const int nst = (int)(w1 - w0);
  // issue cursor
  int it = S ? (int)(w0 / S) : 0, ib = 0, ic = S ? (int)(w0 % S) : 0, inch = 0;
  if (nst > 0) {
    for (;; ++ib) {
      inch = (p.counts[ib] + WP - 1) / WP;
      if (ic < inch) break;
      ic -= inch;
    }
  }
  const int g_row = lane >> 5, g_c4 = lane & 31;   // G / H ops: 2 rows x 128 floats
  const int x_row = lane >> 4, x_c4 = lane & 15;   // X op: 4 rows x 64 floats
  const unsigned x_bytes = (unsigned)min((int64_t)p.B * p.N * 256, (int64_t)0x7fffffff);
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, x_bytes, 0x00020000);
  const unsigned pl_bytes = (unsigned)min((int64_t)p.B * p.N * 512, (int64_t)0x7fffffff);

  auto issue = [&](int buf) {   // DMA the stage at the cursor into ring slot `buf`, then advance the cursor
    float* G = ring + buf * STG;
    float* H = G + GSZ;
    float* X = H + GSZ;
    const int cnt = p.counts[ib];
    const int row0 = ic * WP;                         // first row of the chunk within the sample
    const int64_t srow = (int64_t)ib * p.N + row0;    // global row
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(gplane + it * p.iter_stride), 0, pl_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t hr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(hplane + it * p.iter_stride), 0, pl_bytes, 0x00020000);
    if constexpr (BF) {
      // bf16 planes (written by the bf16-mode forward / backward kernels): a row is 128 bf16 = 256 B at the head of its
      // 512-B slot; one DMA instruction moves 4 rows, the wave's share of the stage is one instruction per plane.
      // LDS tiles G, H: [16 rows][128 bf16]
      const int r = 4 * wave + (lane >> 4);
      const unsigned vo = (row0 + r < cnt) ? (unsigned)((srow + r) * 512 + (lane & 15) * 16) : BAD;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(gr, (lds_ptr_t)(G + wave * 256), 16, vo, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(hr, (lds_ptr_t)(H + wave * 256), 16, vo, 0, 0, 0);
    } else {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int op = 2 * wave + k;                    // rows 2 op, 2 op + 1
        const int r = 2 * op + g_row;
        const unsigned vo = (row0 + r < cnt) ? (unsigned)(((srow + r) * 128 + g_c4 * 4) * 4) : BAD;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(gr, (lds_ptr_t)(G + op * 256), 16, vo, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(hr, (lds_ptr_t)(H + op * 256), 16, vo, 0, 0, 0);
      }
    }
    {
      const int r = 4 * wave + x_row;
      const unsigned vo = (row0 + r < cnt) ? (unsigned)(((srow + r) * 64 + x_c4 * 4) * 4) : BAD;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr_t)(X + wave * 256), 16, vo, 0, 0, 0);
    }
    if (++ic == inch) {
      ic = 0;
      do {
        if (++ib == p.B) { ib = 0; ++it; }
        inch = (p.counts[ib] + WP - 1) / WP;
      } while (inch == 0);
    }
  };

  f32x16 acc[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // per-lane operand offsets inside a stage: a = G[px][wco*64 + 32 i + li]; b_j = column wci*96 + 32 j + li of [H | X]
  const int a_off = kh * 128 + wco * 64 + li;
  int b_off[3], b_pitch[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int ci = wci * 96 + 32 * j;
    if (ci < 128) { b_off[j] = WP * 128 + kh * 128 + ci + li; b_pitch[j] = 256; }
    else { b_off[j] = 2 * WP * 128 + kh * 64 + (ci - 128) + li; b_pitch[j] = 128; }
  }

#pragma unroll
  for (int d = 0; d < WD - 1; ++d)
    if (d < nst) issue(d);
  for (int i = 0; i < nst; ++i) {
    // this wave's DMA for stage i has landed (OPS instructions per stage and wave; the WD - 2 younger stages may be in flight) ...
    switch (min(WD - 2, nst - 1 - i) * OPS) {
      case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
    __syncthreads();   // ... and everyone's; all waves are also done reading slot (i - 1) % WD
    if (i + WD - 1 < nst) issue((i + WD - 1) % WD);
    const float* st = ring + (i % WD) * STG;
    if constexpr (BF) {
      // bf16 operands (mixed-precision training): the stage's 16 rows are ONE k step of v_mfma_f32_32x32x16_bf16 -- lane
      // (column, kh) holds rows 8 kh .. 8 kh + 7
      bf16x8_t a8[2], b8[3];
      const __bf16* g16 = reinterpret_cast<const __bf16*>(st);         // G tile: bf16 [16][128]
      const __bf16* h16 = reinterpret_cast<const __bf16*>(st + GSZ);   // H tile: bf16 [16][128]
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        a8[0][k] = g16[(8 * kh + k) * 128 + wco * 64 + li];
        a8[1][k] = g16[(8 * kh + k) * 128 + wco * 64 + 32 + li];
      }
      // columns of [H | X]: wave column group 0 owns h columns 0..95, group 1 owns h 96..127 and x 0..63 (one wave-uniform
      // branch; a per-element select between the bf16 H tile and the fp32 X tile compiles to a branch per LDS read)
      auto ld_h = [&](int ci) {
        bf16x8_t v;
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = h16[(8 * kh + k) * 128 + ci + li];
        return v;
      };
      auto ld_x = [&](int cx) {
        bf16x8_t v;
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (__bf16)st[2 * GSZ + (8 * kh + k) * 64 + cx + li];
        return v;
      };
      if (wci == 0) {
        b8[0] = ld_h(0); b8[1] = ld_h(32); b8[2] = ld_h(64);
      } else {
        b8[0] = ld_h(96); b8[1] = ld_x(0); b8[2] = ld_x(32);
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[0], b8[j], acc[0][j], 0, 0, 0);
        acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[1], b8[j], acc[1][j], 0, 0, 0);
      }
      continue;
    }
    if constexpr (X2) {
      bf16x8_t ah[2], al[2], bh[3], bl[3];
      auto split8 = [&](const float* col, int pitch, bf16x8_t& hi, bf16x8_t& lo) {   // rows 8 kh .. 8 kh + 7 of one column
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float v = col[(8 * kh + k) * pitch];
          hi[k] = (__bf16)v;
          lo[k] = (__bf16)(v - (float)hi[k]);
        }
      };
      split8(st + wco * 64 + li, 128, ah[0], al[0]);
      split8(st + wco * 64 + 32 + li, 128, ah[1], al[1]);
      if (wci == 0) {   // (wave-uniform, as in the bf16 form)
#pragma unroll
        for (int j = 0; j < 3; ++j) split8(st + WP * 128 + 32 * j + li, 128, bh[j], bl[j]);
      } else {
        split8(st + WP * 128 + 96 + li, 128, bh[0], bl[0]);
        split8(st + 2 * WP * 128 + li, 64, bh[1], bl[1]);
        split8(st + 2 * WP * 128 + 32 + li, 64, bh[2], bl[2]);
      }
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
          f32x16 c = acc[i2][j];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i2], bh[j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i2], bl[j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i2], bh[j], c, 0, 0, 0);
          acc[i2][j] = c;
        }
      continue;
    }
#pragma unroll
    for (int ks = 0; ks < WP / 2; ++ks) {
      const float a0 = st[a_off + ks * 256], a1 = st[a_off + ks * 256 + 32];
      float b[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) b[j] = st[b_off[j] + ks * b_pitch[j]];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b[j], acc[0][j], 0, 0, 0);
        acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b[j], acc[1][j], 0, 0, 0);
      }
    }
  }
  float* o = p.ws + ((int64_t)split * 384 + gate * 128) * 192;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int ci = wci * 96 + 32 * j + li;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = wco * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
        o[co * 192 + ci] = acc[i][j][e];
      }
    }
#endif
}

// ---- weight gradient of the head's first layer, dW1 [32][192] = dpre1^T [hT | x] over the valid rows (round 4) --------------
// Was two generic 1x1 weight-gradient GEMMs on the transposed problem (32 input channels: 0.83 ms per step at 20-50 TFLOP/s,
// 1.5 TB/s) + a transpose.  Here one streaming pass in gru_wgrad_kernel's style: 16-row stages of dpre1 [16][32], hT [16][128] and
// x [16][64] by LDS-DMA through a 3-deep ring; six waves, wave j owns the 32 x 32 block of columns 32 j of [hT | x]; bf16x2
// products (two bf16 planes per operand, three MFMAs) as the gate kernel's X2 form.  Partials ws[split][32][192].
struct GruHeadWgradParams {
  const float* dpre;   // [B*N][32]
  const float* hT;     // [B*N][128]   (plane 5 of the save buffer)
  const float* x;      // [B*N][64]
  const int32_t* counts;
  int B, N, nsplit;
  float* ws;           // [nsplit][32][192]
};

__global__ __launch_bounds__(384) void gru_head_wgrad_kernel(GruHeadWgradParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int STG = WP * (32 + 128 + 64);        // floats per stage: G [16][32] | H [16][128] | X [16][64]  (14 KB)
  constexpr int WD = 3;
  __shared__ __attribute__((aligned(16))) float ring[WD * STG + 256 * 4];   // + 4 KB: landing zone of the spare DMA slots
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kh = lane >> 5;
  const int split = blockIdx.x;
  int S = 0;
  for (int b = 0; b < p.B; ++b) S += (p.counts[b] + WP - 1) / WP;
  const int64_t w0 = (int64_t)S * split / p.nsplit, w1 = (int64_t)S * (split + 1) / p.nsplit;
  const int nst = (int)(w1 - w0);
  int ib = 0, ic = (int)w0, inch = 0;
  if (nst > 0) {
    for (;; ++ib) {
      inch = (p.counts[ib] + WP - 1) / WP;
      if (ic < inch) break;
      ic -= inch;
    }
  }
  const unsigned rows_b = (unsigned)min((int64_t)p.B * p.N, (int64_t)0x3fffff);
  const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dpre), 0, rows_b * 128u, 0x00020000);
  const __amdgpu_buffer_rsrc_t hr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.hT), 0, rows_b * 512u, 0x00020000);
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, rows_b * 256u, 0x00020000);
  // 14 one-KB DMA ops per stage: 0-1 G (8 rows x 128 B each), 2-9 H (2 rows x 512 B), 10-13 X (4 rows x 256 B); op = wave + 6 e,
  // e < 3 -- ops 14 .. 17 land in the spare zone (every wave issues exactly three per stage: compile-time wait counts)
  auto issue = [&](int buf) {
    float* st = ring + buf * STG;
    const int cnt = p.counts[ib];
    const int row0 = ic * WP;
    const int64_t srow = (int64_t)ib * p.N + row0;
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      const int op = wave + 6 * e;
      if (op < 2) {
        const int r = 8 * op + (lane >> 3);
        const unsigned vo = (row0 + r < cnt) ? (unsigned)((srow + r) * 128 + (lane & 7) * 16) : BAD;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(gr, (lds_ptr_t)(st + op * 256), 16, vo, 0, 0, 0);
      } else if (op < 10) {
        const int r = 2 * (op - 2) + (lane >> 5);
        const unsigned vo = (row0 + r < cnt) ? (unsigned)((srow + r) * 512 + (lane & 31) * 16) : BAD;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(hr, (lds_ptr_t)(st + WP * 32 + (op - 2) * 256), 16, vo, 0, 0, 0);
      } else if (op < 14) {
        const int r = 4 * (op - 10) + (lane >> 4);
        const unsigned vo = (row0 + r < cnt) ? (unsigned)((srow + r) * 256 + (lane & 15) * 16) : BAD;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr_t)(st + WP * 160 + (op - 10) * 256), 16, vo, 0, 0, 0);
      } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr_t)(ring + WD * STG + (op - 14) * 256), 16, BAD, 0, 0, 0);
      }
    }
    if (++ic == inch) {
      ic = 0;
      do {
        if (++ib == p.B) { ib = 0; break; }
        inch = (p.counts[ib] + WP - 1) / WP;
      } while (inch == 0);
    }
  };
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
  for (int d = 0; d < WD - 1; ++d)
    if (d < nst) issue(d);
  for (int i = 0; i < nst; ++i) {
    switch (min(WD - 2, nst - 1 - i)) {
      case 1: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
    __syncthreads();
    if (i + WD - 1 < nst) issue((i + WD - 1) % WD);
    const float* st = ring + (i % WD) * STG;
    bf16x8_t ah, al, bh, bl;
    auto split8 = [&](const float* col, int pitch, bf16x8_t& hi, bf16x8_t& lo) {   // rows 8 kh .. 8 kh + 7 of one column
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float v = col[(8 * kh + k) * pitch];
        hi[k] = (__bf16)v;
        lo[k] = (__bf16)(v - (float)hi[k]);
      }
    };
    split8(st + li, 32, ah, al);                                                   // dpre1 column li
    if (wave < 4) split8(st + WP * 32 + 32 * wave + li, 128, bh, bl);              // hT columns 32 wave + li
    else split8(st + WP * 160 + 32 * (wave - 4) + li, 64, bh, bl);                 // x columns
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
  }
  float* o = p.ws + (int64_t)split * 32 * 192;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int co = (e & 3) + 8 * (e >> 2) + 4 * kh;
    o[co * 192 + 32 * wave + li] = acc[e];
  }
#endif
}


// ---- round 5, the "lean" decoder (decoder4.hip): the x columns of every weight gradient come from the [416][4] sums of the backward
// kernel, so the weight-gradient pass multiplies only the 128 h columns: dW_g[:, :128] = dg_pre^T h_in (z, r) / dq_pre^T (r * h).  One
// gate's [128 co x 128 ci] tile per workgroup, waves 2 x 2 (64 x 64 each, four 32 x 32 accumulators), 16-row stages G [16][128] |
// H [16][128] through a four-deep DMA ring (16 KB per stage in fp32).  G planes: gplanes [4][T][B*N][128] = dz_pre | dr_pre | dq_pre |
// r * h; h_in: hsave [T + 1][B*N][128] (the forward's planes).  Partials ws[split][384][128].
struct GruWgrad4Params {
  const float* hsave;
  const float* gplanes;
  const int32_t* counts;
  int B, N, T, nsplit;
  int64_t plane_stride, iter_stride;   // floats
  float* ws;
};

template <int MODE>   // 0: fp32 MFMA, 1: bf16 planes + bf16 MFMA, 3: fp32 planes, bf16x2 products
__global__ __launch_bounds__(256, 2) void gru_wgrad4_kernel(GruWgrad4Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr bool BF = MODE == 1, X2 = MODE == 3;
  constexpr int WD = 4;
  constexpr int GSZ = BF ? WP * 64 : WP * 128;   // floats of the G (and H) tile
  constexpr int STG = 2 * GSZ;
  constexpr int OPS = BF ? 2 : 4;                // DMA instructions per wave and stage
  __shared__ __attribute__((aligned(16))) float ring[WD * STG];   // 64 KB (bf16 planes: 32 KB)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kh = lane >> 5;
  const int wco = wave & 1, wci = wave >> 1;
  const int lg = df_xcd_swizzle(blockIdx.x, gridDim.x);
  const int split = lg / 3, gate = lg - 3 * split;   // gate 0: z, 1: r, 2: q
  const float* gplane = p.gplanes + gate * p.plane_stride;
  const float* hplane = gate == 2 ? p.gplanes + 3 * p.plane_stride : p.hsave;

  int S = 0;
  for (int b = 0; b < p.B; ++b) S += (p.counts[b] + WP - 1) / WP;
  const int64_t total = (int64_t)S * p.T;
  const int64_t w0 = total * split / p.nsplit, w1 = total * (split + 1) / p.nsplit;
  const int nst = (int)(w1 - w0);
  int it = S ? (int)(w0 / S) : 0, ib = 0, ic = S ? (int)(w0 % S) : 0, inch = 0;
  if (nst > 0) {
    for (;; ++ib) {
      inch = (p.counts[ib] + WP - 1) / WP;
      if (ic < inch) break;
      ic -= inch;
    }
  }
  const int g_row = lane >> 5, g_c4 = lane & 31;
  const unsigned pl_bytes = (unsigned)min((int64_t)p.B * p.N * 512, (int64_t)0x7fffffff);

  auto issue = [&](int buf) {
    float* G = ring + buf * STG;
    float* H = G + GSZ;
    const int cnt = p.counts[ib];
    const int row0 = ic * WP;
    const int64_t srow = (int64_t)ib * p.N + row0;
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(gplane + it * p.iter_stride), 0, pl_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t hr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(hplane + it * p.iter_stride), 0, pl_bytes, 0x00020000);
    if constexpr (BF) {
      const int r = 4 * wave + (lane >> 4);
      const unsigned vo = (row0 + r < cnt) ? (unsigned)((srow + r) * 512 + (lane & 15) * 16) : BAD;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(gr, (lds_ptr_t)(G + wave * 256), 16, vo, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(hr, (lds_ptr_t)(H + wave * 256), 16, vo, 0, 0, 0);
    } else {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int op = 2 * wave + k;
        const int r = 2 * op + g_row;
        const unsigned vo = (row0 + r < cnt) ? (unsigned)(((srow + r) * 128 + g_c4 * 4) * 4) : BAD;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(gr, (lds_ptr_t)(G + op * 256), 16, vo, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(hr, (lds_ptr_t)(H + op * 256), 16, vo, 0, 0, 0);
      }
    }
    if (++ic == inch) {
      ic = 0;
      do {
        if (++ib == p.B) { ib = 0; ++it; }
        inch = (p.counts[ib] + WP - 1) / WP;
      } while (inch == 0);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

#pragma unroll
  for (int d = 0; d < WD - 1; ++d)
    if (d < nst) issue(d);
  for (int i = 0; i < nst; ++i) {
    // this wave's DMA for stage i has landed (OPS instructions per stage and wave; up to WD - 2 younger stages in flight) ...
    switch (min(WD - 2, nst - 1 - i) * OPS) {
      case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
    __syncthreads();   // ... and everyone's; all waves are also done reading slot (i - 1) % WD
    if (i + WD - 1 < nst) issue((i + WD - 1) % WD);
    const float* st = ring + (i % WD) * STG;
    if constexpr (BF) {
      bf16x8_t a8[2], b8[2];
      const __bf16* g16 = reinterpret_cast<const __bf16*>(st);
      const __bf16* h16 = reinterpret_cast<const __bf16*>(st + GSZ);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        a8[0][k] = g16[(8 * kh + k) * 128 + wco * 64 + li];
        a8[1][k] = g16[(8 * kh + k) * 128 + wco * 64 + 32 + li];
        b8[0][k] = h16[(8 * kh + k) * 128 + wci * 64 + li];
        b8[1][k] = h16[(8 * kh + k) * 128 + wci * 64 + 32 + li];
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[0], b8[j], acc[0][j], 0, 0, 0);
        acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[1], b8[j], acc[1][j], 0, 0, 0);
      }
      continue;
    }
    if constexpr (X2) {
      bf16x8_t ah[2], al[2], bh[2], bl[2];
      auto split8 = [&](const float* col, bf16x8_t& hi, bf16x8_t& lo) {   // rows 8 kh .. 8 kh + 7 of one column
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float v = col[(8 * kh + k) * 128];
          hi[k] = (__bf16)v;
          lo[k] = (__bf16)(v - (float)hi[k]);
        }
      };
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2) {
        split8(st + wco * 64 + 32 * i2 + li, ah[i2], al[i2]);
        split8(st + GSZ + wci * 64 + 32 * i2 + li, bh[i2], bl[i2]);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
          f32x16 c = acc[i2][j];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i2], bh[j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i2], bl[j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i2], bh[j], c, 0, 0, 0);
          acc[i2][j] = c;
        }
      continue;
    }
#pragma unroll
    for (int ks = 0; ks < WP / 2; ++ks) {
      const float* a = st + (2 * ks + kh) * 128 + wco * 64 + li;
      const float* bq = st + GSZ + (2 * ks + kh) * 128 + wci * 64 + li;
      const float a0 = a[0], a1 = a[32], b0 = bq[0], b1 = bq[32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
  }
  float* o = p.ws + ((int64_t)split * 384 + gate * 128) * 128;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int ci = wci * 64 + 32 * j + li;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = wco * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
        o[co * 128 + ci] = acc[i][j][e];
      }
    }
#endif
}

// the head's dW1[:, :128] = dpre1^T hT (the x columns come from the backward kernel's sums): gru_head_wgrad_kernel without the x
// operand -- four waves, wave j owns columns 32 j of hT; 10 one-KB DMA ops per stage (2 G + 8 H) in 12 slots.  ws[split][32][128].
__global__ __launch_bounds__(256) void gru_head_wgrad4_kernel(GruHeadWgradParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int STG = WP * (32 + 128);
  constexpr int WD = 3;
  __shared__ __attribute__((aligned(16))) float ring[WD * STG + 256 * 2];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kh = lane >> 5;
  const int split = blockIdx.x;
  int S = 0;
  for (int b = 0; b < p.B; ++b) S += (p.counts[b] + WP - 1) / WP;
  const int64_t w0 = (int64_t)S * split / p.nsplit, w1 = (int64_t)S * (split + 1) / p.nsplit;
  const int nst = (int)(w1 - w0);
  int ib = 0, ic = (int)w0, inch = 0;
  if (nst > 0) {
    for (;; ++ib) {
      inch = (p.counts[ib] + WP - 1) / WP;
      if (ic < inch) break;
      ic -= inch;
    }
  }
  const unsigned rows_b = (unsigned)min((int64_t)p.B * p.N, (int64_t)0x3fffff);
  const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dpre), 0, rows_b * 128u, 0x00020000);
  const __amdgpu_buffer_rsrc_t hr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.hT), 0, rows_b * 512u, 0x00020000);
  auto issue = [&](int buf) {
    float* st = ring + buf * STG;
    const int cnt = p.counts[ib];
    const int row0 = ic * WP;
    const int64_t srow = (int64_t)ib * p.N + row0;
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      const int op = wave + 4 * e;
      if (op < 2) {
        const int r = 8 * op + (lane >> 3);
        const unsigned vo = (row0 + r < cnt) ? (unsigned)((srow + r) * 128 + (lane & 7) * 16) : BAD;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(gr, (lds_ptr_t)(st + op * 256), 16, vo, 0, 0, 0);
      } else if (op < 10) {
        const int r = 2 * (op - 2) + (lane >> 5);
        const unsigned vo = (row0 + r < cnt) ? (unsigned)((srow + r) * 512 + (lane & 31) * 16) : BAD;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(hr, (lds_ptr_t)(st + WP * 32 + (op - 2) * 256), 16, vo, 0, 0, 0);
      } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(hr, (lds_ptr_t)(ring + WD * STG + (op - 10) * 256), 16, BAD, 0, 0, 0);
      }
    }
    if (++ic == inch) {
      ic = 0;
      do {
        if (++ib == p.B) { ib = 0; break; }
        inch = (p.counts[ib] + WP - 1) / WP;
      } while (inch == 0);
    }
  };
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
  for (int d = 0; d < WD - 1; ++d)
    if (d < nst) issue(d);
  for (int i = 0; i < nst; ++i) {
    switch (min(WD - 2, nst - 1 - i)) {
      case 1: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
    __syncthreads();
    if (i + WD - 1 < nst) issue((i + WD - 1) % WD);
    const float* st = ring + (i % WD) * STG;
    bf16x8_t ah, al, bh, bl;
    auto split8 = [&](const float* col, int pitch, bf16x8_t& hi, bf16x8_t& lo) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float v = col[(8 * kh + k) * pitch];
        hi[k] = (__bf16)v;
        lo[k] = (__bf16)(v - (float)hi[k]);
      }
    };
    split8(st + li, 32, ah, al);
    split8(st + WP * 32 + 32 * wave + li, 128, bh, bl);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
  }
  float* o = p.ws + (int64_t)split * 32 * 128;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int co = (e & 3) + 8 * (e >> 2) + 4 * kh;
    o[co * 128 + 32 * wave + li] = acc[e];
  }
#endif
}

}  // namespace

// dW1 partials [nsplit][32][192] of the decoder head's first layer from dpre1 [B*N][32], hT [B*N][128] and x [B*N][64] (valid rows
// only); df_conv2d_wgrad_reduce(ws, nsplit, 32, 1, 192, ...) sums them.  bf16x2 products (16 significant bits) like mfma_bf16 = 3.
extern "C" int df_gru_head_wgrad(const float* dpre, const float* hT, const float* x, const int32_t* counts, int B, int N, float* ws,
                                 int nsplit, void* stream) {
  DF_REQUIRE(dpre && hT && x && counts && ws && B > 0 && N > 0 && nsplit >= 1, DF_E_ARG);
  DF_REQUIRE(df_aligned16(dpre) && df_aligned16(hT) && df_aligned16(x), DF_E_ALIGN);
  DF_REQUIRE((int64_t)B * N < (int64_t)0x3fffff, DF_E_SHAPE);      // 32-bit DMA offsets (512 B per hT row)
  GruHeadWgradParams p;
  p.dpre = dpre; p.hT = hT; p.x = x; p.counts = counts; p.B = B; p.N = N; p.nsplit = nsplit; p.ws = ws;
  hipLaunchKernelGGL(gru_head_wgrad_kernel, dim3(nsplit), dim3(384), 0, reinterpret_cast<hipStream_t>(stream), p);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

extern "C" int df_gru_wgrad_splits(void) { return 170; }   // 3 gates x 170 = 510 workgroups: one resident wave at 2 per CU

extern "C" int df_gru_wgrad(const float* save, const float* x, const int32_t* counts, int B, int N, int num_iters,
                            float* ws, int nsplit, void* stream) {
  return df_gru_wgrad_mp(save, x, counts, B, N, num_iters, ws, nsplit, 0, stream);
}

extern "C" int df_gru_wgrad_mp(const float* save, const float* x, const int32_t* counts, int B, int N, int num_iters,
                               float* ws, int nsplit, int mfma_bf16, void* stream) {
  DF_REQUIRE(save && x && counts && ws && B > 0 && N > 0 && num_iters >= 1 && nsplit >= 1, DF_E_ARG);
  DF_REQUIRE(df_aligned16(save) && df_aligned16(x), DF_E_ALIGN);
  DF_REQUIRE((int64_t)B * N * 512 < (int64_t)0x7fffffff, DF_E_SHAPE);  // 32-bit DMA offsets within one iteration's plane
  GruWgradParams p;
  p.save = save; p.x = x; p.counts = counts; p.B = B; p.N = N; p.T = num_iters; p.nsplit = nsplit;
  p.iter_stride = (int64_t)B * N * 128;
  p.plane_stride = p.iter_stride * num_iters;
  p.ws = ws;
  if (mfma_bf16 == 3) hipLaunchKernelGGL((gru_wgrad_kernel<false, true>), dim3(nsplit * 3), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
  else if (mfma_bf16) hipLaunchKernelGGL(gru_wgrad_kernel<true>, dim3(nsplit * 3), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
  else hipLaunchKernelGGL(gru_wgrad_kernel<false>, dim3(nsplit * 3), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

// ---- round 5: the lean decoder's weight-gradient pass (see gru_wgrad4_kernel) ------------------------------------------------------
extern "C" int df_gru_lean_wgrad(const float* hsave, const float* gplanes, const int32_t* counts, int B, int N, int num_iters, float* ws,
                                 int nsplit, int mfma_bf16, void* stream) {
  DF_REQUIRE(hsave && gplanes && counts && ws && B > 0 && N > 0 && num_iters >= 1 && nsplit >= 1, DF_E_ARG);
  DF_REQUIRE(df_aligned16(hsave) && df_aligned16(gplanes), DF_E_ALIGN);
  DF_REQUIRE((int64_t)B * N * 512 < (int64_t)0x7fffffff, DF_E_SHAPE);  // 32-bit DMA offsets within one iteration's plane
  GruWgrad4Params p;
  p.hsave = hsave; p.gplanes = gplanes; p.counts = counts; p.B = B; p.N = N; p.T = num_iters; p.nsplit = nsplit;
  p.iter_stride = (int64_t)B * N * 128;
  p.plane_stride = p.iter_stride * num_iters;
  p.ws = ws;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (mfma_bf16 == 3) hipLaunchKernelGGL(gru_wgrad4_kernel<3>, dim3(nsplit * 3), dim3(256), 0, s, p);
  else if (mfma_bf16) hipLaunchKernelGGL(gru_wgrad4_kernel<1>, dim3(nsplit * 3), dim3(256), 0, s, p);
  else hipLaunchKernelGGL(gru_wgrad4_kernel<0>, dim3(nsplit * 3), dim3(256), 0, s, p);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

extern "C" int df_gru_lean_head_wgrad(const float* dpre, const float* hT, const int32_t* counts, int B, int N, float* ws, int nsplit,
                                      void* stream) {
  DF_REQUIRE(dpre && hT && counts && ws && B > 0 && N > 0 && nsplit >= 1, DF_E_ARG);
  DF_REQUIRE(df_aligned16(dpre) && df_aligned16(hT), DF_E_ALIGN);
  DF_REQUIRE((int64_t)B * N < (int64_t)0x3fffff, DF_E_SHAPE);
  GruHeadWgradParams p;
  p.dpre = dpre; p.hT = hT; p.x = nullptr; p.counts = counts; p.B = B; p.N = N; p.nsplit = nsplit; p.ws = ws;
  hipLaunchKernelGGL(gru_head_wgrad4_kernel, dim3(nsplit), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
  DF_CHECK_LAUNCH();
  return DF_OK;
}
