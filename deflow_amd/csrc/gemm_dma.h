// Weight-streaming GEMM of the point-decoder kernels, LDS-DMA form (see decoder3.hip for the design notes):
// a workgroup of 4 waves multiplies each wave's private 16-row A operand by weight rows streamed L2 -> LDS in 32-deep
// k chunks with buffer_load ... lds, double buffered and one chunk ahead -- also across consecutive GEMMs.
// v_mfma_f32_16x16x4_f32 with the k permutation k = 16 g + 4 (lane >> 4) + s, so every fragment is one ds_read_b128.
#pragma once
#include "common.h"

namespace gd {

constexpr int LDH = 132;            // A region pitch (floats): 33 slots of 16 B -> conflict-free b128 rows
constexpr int BT = 128 * 32;        // one weight buffer: 128 rows x 32 floats, unpadded
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
using rsrc_t = __amdgpu_buffer_rsrc_t;

// `base` and `bytes` must be wave-uniform (they are: per-wave row blocks).  The explicit readfirstlane matters: 64-bit
// address arithmetic such as (int64) b * N + row is selected as VALU (v_mad_u64_u32), the descriptor then lives in VGPRs,
// and EVERY buffer instruction using it gets a "waterfall" loop (4 readfirstlane + 2 v_cmp_eq_u64 + saveexec + branch) --
// 188 of them in gru_bwd3_kernel, 12 instructions per plane load or store instead of one.
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  void* ub = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(ub, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
// DF_BUF_ST_AUX (compile time; A/B through build.build_variant): cache policy of the plane stores -- 0 default, 2 = nt (streaming)
#ifndef DF_BUF_ST_AUX
#define DF_BUF_ST_AUX 0
#endif
__device__ __forceinline__ void buf_st4(rsrc_t r, unsigned voff, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, 0, DF_BUF_ST_AUX);
}
__device__ __forceinline__ void buf_st1(rsrc_t r, unsigned voff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, 0, 0);
}
// bf16 saved planes of the mixed-precision GRU (planes 0..4 of the save buffer in bf16 mode): a row keeps its 512-byte slot and
// holds 128 bf16 in the first 256 bytes, so plane / iteration / row offsets are those of the fp32 layout.
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned short bf16_bits(float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); }
__device__ __forceinline__ float bf16_float(unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); }
__device__ __forceinline__ void buf_st4_bf16(rsrc_t r, unsigned voff, f32x4 v) {   // 4 floats -> 4 bf16 (8 bytes)
  bf16x2_t a, b;
  a[0] = (__bf16)v[0]; a[1] = (__bf16)v[1]; b[0] = (__bf16)v[2]; b[1] = (__bf16)v[3];
  u32x2_t w;
  w[0] = __builtin_bit_cast(unsigned, a);
  w[1] = __builtin_bit_cast(unsigned, b);
  __builtin_amdgcn_raw_buffer_store_b64(w, r, voff, 0, 0);
}
__device__ __forceinline__ void buf_st1_bf16(rsrc_t r, unsigned voff, float v) {
  __builtin_amdgcn_raw_buffer_store_b16(bf16_bits(v), r, voff, 0, 0);
}
__device__ __forceinline__ float buf_ld1_bf16(rsrc_t r, unsigned voff) {
  return bf16_float(__builtin_amdgcn_raw_buffer_load_b16(r, voff, 0, 0));
}

// 32-deep k chunk `chunk` of ROWS weight rows (LDW floats apart) -> LDS buffer, all four waves cooperating: one DMA
// instruction moves 8 rows x 128 B, wave w takes row groups w, w + 4, ...  voff = WStream::voff<LDW>.
// NWV = waves of the workgroup sharing the weight stream (4; 8 in the lean decoder's 128-point form, round 6)
template <int ROWS, int LDW, int NWV = 4>
__device__ __forceinline__ void dma_chunk(const float* __restrict__ W, int chunk, float* Bbuf, int wave, unsigned voff) {
  const rsrc_t r = make_rsrc(W, 0x7fffffffu);
#pragma unroll
  for (int i = 0; i < (ROWS + 8 * NWV - 1) / (8 * NWV); ++i)
    if (NWV == 4 || wave * 8 + 8 * NWV * i < ROWS)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(Bbuf + (wave + NWV * i) * 256), 16, voff,
                                               (unsigned)((i * 8 * NWV * LDW + chunk * 32) * 4), 0, 0);
}

// The same for PRE-CAST bf16 weights (W points at bf16 data, LDW in elements): a row of the chunk is 64 B, one DMA instruction
// moves 16 rows, the four waves cover 64 rows per pass.  voff = WStream::voff16<LDW>.
template <int ROWS, int LDW, int NWV = 4>
__device__ __forceinline__ void dma_chunk16(const float* __restrict__ W, int chunk, float* Bbuf, int wave, unsigned voff) {
  const rsrc_t r = make_rsrc(W, 0x7fffffffu);
#pragma unroll
  for (int i = 0; i < (ROWS + 16 * NWV - 1) / (16 * NWV); ++i)
    if (wave * 16 + 16 * NWV * i < ROWS)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(Bbuf + (wave + NWV * i) * 256), 16, voff,
                                               (unsigned)((i * 16 * NWV * LDW + chunk * 32) * 2), 0, 0);
}

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8_t pack_bf16(const f32x4 lo, const f32x4 hi) {
  bf16x8_t r;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    r[k] = (__bf16)lo[k];
    r[4 + k] = (__bf16)hi[k];
  }
  return r;
}

// BF_ = bf16-operand mode (mixed-precision training, optim.Trainer(dtype="bf16")): a 32-deep k chunk -- the two 16-wide
// k groups a lane reads as two float4 -- is rounded to bf16 (RNE) and goes through ONE v_mfma_f32_16x16x32_bf16 instead of
// eight v_mfma_f32_16x16x4_f32; state, gates, accumulators and the saved planes stay fp32.
// MODE 2 = MODE 1 with the weights pre-cast to bf16 in memory (one cast per optimizer step): the chunk tile is [rows][64 B]
// (16-byte slots swizzled with (row >> 2) & 3), a B fragment is ONE ds_read_b128 of 8 consecutive k -- the fp32-tile form reads
// two b128 and converts 8 values per fragment and is LDS-bound (16 points per wave: every wave re-reads every weight fragment).
// The A operand then holds k = 8 lq .. 8 lq + 7 of the chunk (callers pass a_lane = row + 8 lq and load the x registers to match).
// MODE 3 ("bf16x2", fp32 training; round 3): BOTH operands as two bf16 planes, hi = bf16(v), lo = bf16(v - hi) -- 16 significant
// bits each -- and three MFMAs per product (lo hi', hi lo', hi hi'; the lo lo' term, <= 2^-16 of the product, is dropped): the
// gate GEMMs at 16 / 3 of the fp32-MFMA rate with a relative error <= ~2^-15 per product (rms ~1e-5), which averages out over
// the 128..192-deep sums (measured against float64: tests/test_gpu_kernels.py::test_gru_decoder_golden).  No scales: bf16
// has the fp32 exponent range.  The weights arrive pre-split per optimizer step as rows of [hi (LDW) | lo (LDW)] bf16 -- the
// byte pitch of the fp32 row, so ROW offsets in floats are those of the fp32 layout and COLUMN offsets halve -- and go to LDS as
// two 8 KB tiles per chunk (one weight buffer, exactly); the A operand is split in registers once per chunk.
template <int MODE, int NWV_ = 4>
struct WStreamT {   // the weight-chunk pipeline state shared by consecutive GEMMs
  static constexpr int NWV = NWV_;
  static constexpr bool BF = MODE != 0;
  static constexpr bool W16 = MODE >= 2;
  static constexpr bool X2 = MODE == 3;
  static constexpr int A2 = W16 ? 4 : 16;   // float offset of a lane's second 4-float k group inside a 32-deep chunk
  float* Bs;
  int par, wave;
  unsigned vrow, vslot; // per-lane DMA source: row within the first 32 (wave * 8 + lane / 8), swizzled slot byte offset
  template <int LDW>
  __device__ __forceinline__ unsigned voff() const { return vrow * (unsigned)(LDW * 4) + vslot; }
  unsigned vrow16, vslot16;   // bf16 weights: row within the first 64 (wave * 16 + lane / 4), swizzled slot byte offset
  template <int LDW>
  __device__ __forceinline__ unsigned voff16() const { return vrow16 * (unsigned)(LDW * 2) + vslot16; }
  const float* b_lane; // fragment base: row li of buffer 0
  int bsl[2];          // swizzled slot offsets (floats) of the two 16-wide k groups
  const float* b_lane16;  // bf16 tiles: row li (16 floats = 64 B per row) + this lane's swizzled slot
};
using WStream = WStreamT<0>;

// MODE 3: chunk `chunk` of both planes of ROWS pre-split weight rows ([hi (LDW) | lo (LDW)] bf16 per row) -> the two halves of
// one weight buffer
template <int ROWS, int LDW, class WS>
__device__ __forceinline__ void dma_chunk_x2(const float* __restrict__ W, int chunk, float* Bbuf, const WS& ws) {
  static_assert(ROWS <= 128, "two 8 KB plane tiles per weight buffer");
  dma_chunk16<ROWS, 2 * LDW, WS::NWV>(W, chunk, Bbuf, ws.wave, ws.template voff16<2 * LDW>());
  dma_chunk16<ROWS, 2 * LDW, WS::NWV>(W + LDW / 2, chunk, Bbuf + BT / 2, ws.wave, ws.template voff16<2 * LDW>());
}
// the first chunk of a kernel's first GEMM, in the stream's mode
template <int ROWS, int LDW, class WS>
__device__ __forceinline__ void dma_first(const float* __restrict__ W, int chunk, float* Bbuf, const WS& ws) {
  if constexpr (WS::X2) dma_chunk_x2<ROWS, LDW>(W, chunk, Bbuf, ws);
  else if constexpr (WS::W16) dma_chunk16<ROWS, LDW, WS::NWV>(W, chunk, Bbuf, ws.wave, ws.template voff16<LDW>());
  else dma_chunk<ROWS, LDW, WS::NWV>(W, chunk, Bbuf, ws.wave, ws.template voff<LDW>());
}

// acc[t] += A[16, 32 NCH] * W[ROWS, chunks c0 .. c0 + NCH)^T.  A fragments: LDS (a_lane, chunk c at +32 c) or the x
// registers (XA).  Precondition: chunk c0 is in buffer ws.par, barrier passed.  The first chunk of the next GEMM
// (Wn, cn, ROWS_NEXT rows) is fetched during the last chunk.
template <int ROWS, int NCH, bool XA, int ROWS_NEXT, int LDW = 192, int LDW_NEXT = 192, class WS>
__device__ __forceinline__ void gemm(const float* __restrict__ W, int c0, const float* __restrict__ Wn, int cn,
                                     const float* a_lane, const f32x4 (&xf)[4], WS& ws, f32x4 (&acc)[ROWS / 16]) {
  constexpr int NPAIR = ROWS / 32;
  auto chunk = [&](int c, const f32x4 a0, const f32x4 a1) {
    float* nb = ws.Bs + ((ws.par + c + 1) & 1) * BT;
    if constexpr (WS::X2) {
      if (c + 1 < NCH) dma_chunk_x2<ROWS, LDW>(W, c0 + c + 1, nb, ws);
      else if (Wn) dma_chunk_x2<ROWS_NEXT, LDW_NEXT>(Wn, cn, nb, ws);
      const float* bb = ws.b_lane16 + ((ws.par + c) & 1) * BT;
      bf16x8_t ah, al;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        ah[k] = (__bf16)a0[k];
        al[k] = (__bf16)(a0[k] - (float)ah[k]);
        ah[4 + k] = (__bf16)a1[k];
        al[4 + k] = (__bf16)(a1[k] - (float)ah[4 + k]);
      }
      // (round 5: issuing the three products tile-group-wise -- four independent accumulators between two links of a tile's dependent
      // chain -- left the forward unchanged (2.33 -> 2.29 ms) and cost the backward registers (16 -> 27 spills, 4.88 -> 5.27 ms): the
      // kernels are bound by the per-chunk workgroup barrier, not by the MFMA result latency)
#pragma unroll
      for (int t = 0; t < ROWS / 16; ++t) {
        const bf16x8_t bh = *reinterpret_cast<const bf16x8_t*>(bb + t * 256), bl = *reinterpret_cast<const bf16x8_t*>(bb + BT / 2 + t * 256);
        f32x4 v = acc[t];
        v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, v, 0, 0, 0);   // small terms first
        v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, v, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, v, 0, 0, 0);
        acc[t] = v;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      return;
    }
    if constexpr (WS::W16) {
      if (c + 1 < NCH) dma_chunk16<ROWS, LDW, WS::NWV>(W, c0 + c + 1, nb, ws.wave, ws.template voff16<LDW>());
      else if (Wn) dma_chunk16<ROWS_NEXT, LDW_NEXT, WS::NWV>(Wn, cn, nb, ws.wave, ws.template voff16<LDW_NEXT>());
      const float* bb = ws.b_lane16 + ((ws.par + c) & 1) * BT;
      const bf16x8_t a8 = pack_bf16(a0, a1);
#pragma unroll
      for (int t = 0; t < ROWS / 16; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, *reinterpret_cast<const bf16x8_t*>(bb + t * 256), acc[t], 0, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      return;
    }
    if (c + 1 < NCH) dma_chunk<ROWS, LDW, WS::NWV>(W, c0 + c + 1, nb, ws.wave, ws.template voff<LDW>());
    else if (Wn) dma_chunk<ROWS_NEXT, LDW_NEXT, WS::NWV>(Wn, cn, nb, ws.wave, ws.template voff<LDW_NEXT>());
    const float* bb = ws.b_lane + ((ws.par + c) & 1) * BT;
    if constexpr (WS::BF) {
      const bf16x8_t a8 = pack_bf16(a0, a1);
#pragma unroll
      for (int t = 0; t < ROWS / 16; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, pack_bf16(ld4(bb + t * 512 + ws.bsl[0]), ld4(bb + t * 512 + ws.bsl[1])),
                                                         acc[t], 0, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      return;
    }
    f32x4 nb0 = ld4(bb + ws.bsl[0]), nb1 = ld4(bb + 512 + ws.bsl[0]);
#pragma unroll
    for (int j = 0; j < 2 * NPAIR; ++j) {
      const int g = j / NPAIR, t = 2 * (j % NPAIR);
      const f32x4 b0 = nb0, b1 = nb1;
      if (j + 1 < 2 * NPAIR) {
        const int gn = (j + 1) / NPAIR, tn = 2 * ((j + 1) % NPAIR);
        nb0 = ld4(bb + tn * 512 + ws.bsl[gn]);
        nb1 = ld4(bb + (tn + 1) * 512 + ws.bsl[gn]);
      }
      const f32x4 a = g ? a1 : a0;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b0[s], acc[t], 0, 0, 0);
        acc[t + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b1[s], acc[t + 1], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the next chunk has landed
    __syncthreads();                                     // ... and everyone's; the current buffer is free again
  };
  if constexpr (XA) {
    static_assert(NCH == 2, "the x operand is two chunks");
    chunk(0, xf[0], xf[1]);
    chunk(1, xf[2], xf[3]);
  } else {
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) chunk(c, ld4(a_lane + c * 32), ld4(a_lane + c * 32 + WS::A2));
  }
  ws.par = (ws.par + NCH) & 1;
}

template <class WS>
__device__ __forceinline__ void wstream_init(WS& ws, float* Bs) {
  const int lane = threadIdx.x & 63, li = lane & 15, lq = lane >> 4;
  ws.Bs = Bs;
  ws.par = 0;
  ws.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l3 = lane >> 3, c4 = lane & 7;
  const int row = ws.wave * 8 + l3;                 // (row >> 1) & 7 is the same for row + 32 i
  ws.vrow = (unsigned)row;
  ws.vslot = (unsigned)((c4 ^ ((row >> 1) & 7)) * 16);
  ws.b_lane = Bs + li * 32;
  ws.bsl[0] = ((lq) ^ ((li >> 1) & 7)) * 4;
  ws.bsl[1] = ((4 + lq) ^ ((li >> 1) & 7)) * 4;
  // bf16 weight tiles: DMA lane -> (row wave * 16 + lane / 4, physical slot lane & 3 holds logical slot ^ g((row >> 2) & 3));
  // fragment of lane (li, lq) in row block t: row 16 t + li, logical slot lq.
  // g = (0, 2, 3, 1) (round 5): ds_read_b128 is serviced in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, {32-35, 44-47,
  // 52-59}, {36-43, 48-51, 60-63} (MI355X_MICROARCH.md, LDS), NOT in runs of 16 lanes.  A group holds, per residue li mod 4, the rows
  // (c, c + 12) of one k slot and (c + 4, c + 8) of its neighbour: with the plain swizzle g(x) = x those four 16-byte reads fell on two
  // bank quads -- every weight fragment read took twice its LDS cycles (SQ_LDS_BANK_CONFLICT = 42 % of SQ_LDS_IDX_ACTIVE in
  // gru_fwd4_kernel, profiles/r05_pmc_gru4.txt).  With g the four physical slots {g(0), g(3), 1 ^ g(1), 1 ^ g(2)} (and the three
  // other groups' sets) are permutations of 0..3.
  auto g4 = [](int x) { return (0x78 >> (2 * x)) & 3; };
  const int r16 = lane >> 2;
  ws.vrow16 = (unsigned)(ws.wave * 16 + r16);
  ws.vslot16 = (unsigned)(((lane & 3) ^ g4((r16 >> 2) & 3)) * 16);
  ws.b_lane16 = Bs + li * 16 + ((lq ^ g4((li >> 2) & 3)) * 4);
}

}  // namespace gd
