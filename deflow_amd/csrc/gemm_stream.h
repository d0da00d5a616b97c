// Weight-streaming GEMM for the per-point decoder kernels (forward and backward).
//
// A workgroup of 4 waves multiplies each wave's private A operand (16 rows, resident in LDS) by a weight
// matrix W[ROWS, K] that is streamed L2 -> registers -> LDS in 32-deep k chunks, double buffered and one
// chunk ahead -- also across consecutive GEMMs (the last chunk of one GEMM prefetches the first chunk of
// the next).  v_mfma_f32_16x16x4_f32; k permutation k = 16g + 4*(lane>>4) + s so every fragment fetch is
// one ds_read_b128.
#pragma once
#include "common.h"

namespace gs {

constexpr int LDB = 36;          // B tile pitch (floats)
constexpr int BROWS = 256;       // B tile rows (max)
constexpr int BSZ = BROWS * LDB; // one B buffer (floats)

struct Stager {
  f32x4 r[8];
};

template <int ROWS>
__device__ __forceinline__ void stage_load(Stager& s, const float* __restrict__ W, int ldw, int chunk) {
  const int c4 = threadIdx.x & 7, r0 = threadIdx.x >> 3;
#pragma unroll
  for (int i = 0; i < ROWS / 32; ++i) s.r[i] = ld4(W + (int64_t)(r0 + 32 * i) * ldw + chunk * 32 + c4 * 4);
}
template <int ROWS>
__device__ __forceinline__ void stage_store(const Stager& s, float* Bbuf) {
  const int c4 = threadIdx.x & 7, r0 = threadIdx.x >> 3;
#pragma unroll
  for (int i = 0; i < ROWS / 32; ++i) st4(Bbuf + (r0 + 32 * i) * LDB + c4 * 4, s.r[i]);
}

// acc[t] (16 x 16 tile t of the wave's 16 x ROWS output) += A[16, 32*nchunks] * W[ROWS, 32*nchunks]^T.
// Precondition: chunk 0 of W is in B buffer `par` and a barrier has been passed.  On return the first chunk
// of Wnext (if any) is in buffer `par` (updated) and a barrier has been passed.
template <int ROWS, int ROWS_NEXT, int BS = BSZ>
__device__ __forceinline__ void gemm_stream(const float* __restrict__ W, int ldw, int nchunks,
                                            const float* __restrict__ Wnext, int ldw_next, const float* a_lane,
                                            float* Bs, int& par, f32x4 (&acc)[ROWS / 16], Stager& stg) {
  const int lane = threadIdx.x & 63;
  const float* b_lane = Bs + (lane & 15) * LDB + (lane >> 4) * 4;
  for (int c = 0; c < nchunks; ++c) {
    const bool more = c + 1 < nchunks;
    if (more) stage_load<ROWS>(stg, W, ldw, c + 1);
    else if (Wnext) stage_load<ROWS_NEXT>(stg, Wnext, ldw_next, 0);
    const float* bb = b_lane + ((par + c) & 1) * BS;
    // Fragment reads are software-pipelined one tile pair ahead: with one wave per SIMD nothing else hides the LDS
    // latency, so the reads for pair j+1 are issued before the 8 MFMAs of pair j.  Two accumulators alternate, so
    // back-to-back MFMAs never depend on each other (16x16x4 f32: 32-cycle issue, 40-cycle dependent latency).
    constexpr int NPAIR = ROWS / 32;  // tile pairs per 16-wide k group
    const f32x4 a0 = ld4(a_lane + c * 32), a1 = ld4(a_lane + c * 32 + 16);
    f32x4 nb0 = ld4(bb), nb1 = ld4(bb + 16 * LDB);
#pragma unroll
    for (int j = 0; j < 2 * NPAIR; ++j) {
      const int g = j / NPAIR, t = 2 * (j % NPAIR);
      const f32x4 b0 = nb0, b1 = nb1;
      if (j + 1 < 2 * NPAIR) {
        const int gn = (j + 1) / NPAIR, tn = 2 * ((j + 1) % NPAIR);
        nb0 = ld4(bb + tn * 16 * LDB + gn * 16);
        nb1 = ld4(bb + (tn + 1) * 16 * LDB + gn * 16);
      }
      const f32x4 a = g ? a1 : a0;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b0[s], acc[t], 0, 0, 0);
        acc[t + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b1[s], acc[t + 1], 0, 0, 0);
      }
    }
    float* nb = Bs + ((par + c + 1) & 1) * BS;
    if (more) stage_store<ROWS>(stg, nb);
    else if (Wnext) stage_store<ROWS_NEXT>(stg, nb);
    __syncthreads();
  }
  par = (par + nchunks) & 1;
}


}  // namespace gs
