// Shared device/host helpers for the gfx950 kernels.  CDNA4 only: wave = 64, MFMA f32 32x32x2 / 16x16x4.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

#include "../../include/deflow_amd.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define DF_CHECK_LAUNCH()                      \
  do {                                         \
    hipError_t e__ = hipGetLastError();        \
    if (e__ != hipSuccess) return (int)e__;    \
  } while (0)

#define DF_REQUIRE(cond, code) \
  do {                         \
    if (!(cond)) return (code); \
  } while (0)

// One-time raise of a kernel's dynamic-LDS limit, safe when several host threads make their first call at once (the
// header promises "thread-safe per stream"): std::call_once instead of a racy `static bool`.
#define DF_SET_LDS_ONCE(kernel, bytes)                                                                            \
  do {                                                                                                            \
    static std::once_flag once__;                                                                                 \
    static int rc__ = 0;                                                                                          \
    const int bytes__ = (int)(bytes);                                                                             \
    std::call_once(once__, [bytes__] {                                                                            \
      rc__ = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),                                      \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, bytes__);                        \
    });                                                                                                           \
    if (rc__ != 0) return rc__;                                                                                   \
  } while (0)

static inline bool df_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// element offset of image n inside an image set (see include/deflow_amd.h)
__device__ __forceinline__ int64_t df_img_base(const df_img& d, int n) {
  return (int64_t)(n % d.grp_size) * d.img_stride + (int64_t)(n / d.grp_size) * d.grp_off;
}

// XCD-aware, bijective remap of a 1-D grid: block b runs on XCD b % 8 (observed, speed only);
// give every XCD one contiguous range of logical tiles so neighbouring tiles share an L2.
__device__ __forceinline__ int df_xcd_swizzle(int bid, int nwg) {
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + (bid >> 3);
}

// Index decode of the streaming kernels: every extent of this model (channels / 4, H W, W, H) is a power of two, and a 64-bit division
// by a run-time value is ~25-100 VALU instructions -- a quarter of what bn_gelu_apply did per four elements (round 4).  sh =
// df_pow2_shift(d) once per kernel (scalar), then a shift where d is a power of two and the division elsewhere (uniform branch).
__device__ __forceinline__ int df_pow2_shift(int d) { return (d > 0 && (d & (d - 1)) == 0) ? __builtin_ctz((unsigned)d) : -1; }
__device__ __forceinline__ int64_t df_udiv(int64_t x, int d, int sh) { return sh >= 0 ? (x >> sh) : x / d; }

// exact-erf GELU [REF decoder.py:209] and its derivative.  erf through Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 absolute,
// i.e. ~2 ulp of libm's erff at fp32): one v_exp_f32 -- exp(-x^2/2), shared by the erfc tail and the Gaussian of the
// derivative -- one v_rcp_f32 and a degree-5 Horner form, ~16 VALU instructions instead of libm's erff + expf (~70 with
// their range reductions).  Round 3: with bf16 tensors the BatchNorm + GELU passes stopped being HBM-bound -- half the bytes
// left the time unchanged (2.9 TB/s) -- and the conv epilogues pay the same math per output element.
// -DDF_GELU_LIBM restores the libm forms (A/B).
__device__ __forceinline__ void df_gelu_parts(float x, float& cdf, float& gauss /* exp(-x^2/2) */) {
  gauss = __builtin_amdgcn_exp2f(-0.72134752044448170368f * x * x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.23164189f, fabsf(x), 1.0f));   // 1 / (1 + p |x| / sqrt 2), p = 0.3275911
  float q = fmaf(t, 1.061405429f, -1.453152027f);
  q = fmaf(q, t, 1.421413741f);
  q = fmaf(q, t, -0.284496736f);
  q = fmaf(q, t, 0.254829592f);
  const float half_erfc = 0.5f * q * t * gauss;                               // 0.5 erfc(|x| / sqrt 2)
  cdf = x >= 0.f ? 1.0f - half_erfc : half_erfc;
}
#if defined(DF_GELU_LIBM)
__device__ __forceinline__ float df_gelu(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float df_gelu_grad(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
  return cdf + x * pdf;
}
#else
__device__ __forceinline__ float df_gelu(float x) {
  float cdf, g;
  df_gelu_parts(x, cdf, g);
  return x * cdf;
}
__device__ __forceinline__ float df_gelu_grad(float x) {
  float cdf, g;
  df_gelu_parts(x, cdf, g);
  return fmaf(x * 0.39894228040143267794f, g, cdf);
}
#endif
__device__ __forceinline__ float df_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
// Fast forms for the GRU gates (v_exp_f32 + v_rcp_f32, ~1e-7 relative): the decoder kernels run one wave per SIMD,
// so gate math is not hidden behind another wave's MFMAs and the libm range-reduction versions cost ~30 % of the kernel.
// v_rcp_f32 (1 ulp) on purpose: __frcp_rn / a plain division expand to the 11-instruction correctly-rounded sequence
// (div_scale, rcp, 5 fma, div_fmas, div_fixup) -- 14 VALU instructions per gate value instead of 4, and the GRU kernels
// evaluate 3 x 128 of them per point and iteration
__device__ __forceinline__ float df_sigmoid_fast(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * x));
}
__device__ __forceinline__ float df_tanh_fast(float x) {
  // tanh(x) = 1 - 2 / (1 + e^{2x}); saturates correctly for large |x| (exp2 -> inf or 0)
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008177792681472f * x));
}

// max |x| bookkeeping of the fp16x2 convolution kernels (conv.hip): block-wide max of one non-negative float per thread (blocks of
// up to 1024 threads), then ONE integer atomic max of its bit pattern into *amax -- skipped when the value already there is not smaller (non-
// negative floats order like their bit patterns; the result is exact and order-independent)
__device__ __forceinline__ void df_block_amax(float mf, unsigned* __restrict__ amax) {
  unsigned m = __builtin_bit_cast(unsigned, mf);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
  __shared__ unsigned df_amax_red[16];
  if ((threadIdx.x & 63) == 0) df_amax_red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nw = (int)((blockDim.x + 63) >> 6);
    for (int w = 1; w < nw; ++w) m = max(m, df_amax_red[w]);
    if (m > __atomic_load_n(amax, __ATOMIC_RELAXED)) atomicMax(amax, m);
  }
}
__device__ __forceinline__ float df_amax4(float m, const f32x4 v) {
  return fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
}
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
