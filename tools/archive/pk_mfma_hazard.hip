// Minimal reproducer attempt (round 4) for the finding of tools/pfn_race_probe*.sh: kernels containing packed-fp32 VALU instructions
// return wrong sums while ANOTHER kernel (another stream is enough) executes v_mfma_f32_16x16x32_bf16 on the same chip.
// victim:    every lane keeps a running c = fma(a, b, c) twice -- once with v_pk_fma_f32 on a register pair, once with two v_fma_f32 --
//            and counts the iterations in which the two disagree (they are the same IEEE operation: 0 expected).
// neighbour: a register-resident MFMA stream (selectable instruction) on a second stream.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/pk_mfma_hazard tools/pk_mfma_hazard.hip && tools/bin/pk_mfma_hazard
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void victim(unsigned* bad, float* sink, int iters, int use_lds) {
  __shared__ float lds[256 * 8];
  const int t = blockIdx.x * 256 + threadIdx.x;
  f32x2 a = {1.0f + (t & 1023) * 1e-3f, 0.5f + (t & 511) * 2e-3f}, b = {0.999f, 1.001f}, c = {0.f, 0.f};
  float c0 = 0.f, c1 = 0.f;
  unsigned n = 0;
  for (int i = 0; i < iters; ++i) {
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(c0) : "v"(a.x), "v"(b.x));
    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(c1) : "v"(a.y), "v"(b.y));
    if (use_lds) {      // the pillar kernels' pattern: values through LDS, summed by a few lanes
      lds[(i & 7) * 256 + threadIdx.x] = c.x;
      __syncthreads();
      if (threadIdx.x < 8) c0 += lds[(i & 7) * 256 + threadIdx.x + 8] * 0.f;
    }
    // (float compares of copies: bit_cast<unsigned>(c.y) after the asm compared c.x twice -- the element-extract problem DESIGN.md
    // lists under "found on the way")
    const float cx = c[0], cy = c[1];
    n += (cx != c0 || cy != c1) ? 1u : 0u;
    if ((i & 63) == 63) { c = c * 1e-3f; c0 = c.x; c1 = c.y; }   // keep the values in range; resynchronise the two copies
  }
  if (n) atomicAdd(bad, 1u);        // lanes with at least one mismatch
  if (sink && c.x == 12345.f) sink[0] = c0 + c1;
}

template <int KIND>
__global__ __launch_bounds__(256) void neighbour(const float* __restrict__ ops, float* __restrict__ out, int iters) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  float s = 0.f;
  if constexpr (KIND == 0) {          // v_mfma_f32_16x16x32_bf16
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)ops[(t * 8 + e) & 65535]; b[e] = (__bf16)ops[(t * 8 + e + 4096) & 65535]; }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, b, c3, 0, 0, 0);
    }
    s = c0[0] + c1[1] + c2[2] + c3[3];
  } else if constexpr (KIND == 1) {   // v_mfma_f32_32x32x16_bf16
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)ops[(t * 8 + e) & 65535]; b[e] = (__bf16)ops[(t * 8 + e + 4096) & 65535]; }
    f32x16 c0, c1;
    for (int e = 0; e < 16; ++e) { c0[e] = 0.f; c1[e] = 0.f; }
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c1, 0, 0, 0);
    }
    s = c0[0] + c1[1];
  } else if constexpr (KIND == 2) {   // v_mfma_f32_16x16x4_f32
    const float a = ops[t & 65535], b = ops[(t + 77) & 65535];
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, a, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, b, c3, 0, 0, 0);
    }
    s = c0[0] + c1[1] + c2[2] + c3[3];
  } else {                            // v_mfma_f32_16x16x32_f16
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)ops[(t * 8 + e) & 65535]; b[e] = (_Float16)ops[(t * 8 + e + 4096) & 65535]; }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, a, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, b, c3, 0, 0, 0);
    }
    s = c0[0] + c1[1] + c2[2] + c3[3];
  }
  out[t] = s;
}

// the neighbour alone, for tools/pfn_bwd_stress.py (DF_STRESS_SIDE=mfma:<kind>): built with -DHAZARD_LIB -shared into tools/bin/libhazard.so
extern "C" int hazard_neighbour_launch(void* stream, int kind, int iters, int blocks) {
  static float *ops = nullptr, *out = nullptr;
  if (!ops) {
    hipMalloc(&ops, 65536 * 4);
    hipMalloc(&out, 4096 * 256 * 4);
    float* h = (float*)malloc(65536 * 4);
    srand(3);
    for (int i = 0; i < 65536; ++i) h[i] = (rand() / (float)RAND_MAX * 2.f - 1.f) * 0.25f;
    hipMemcpy(ops, h, 65536 * 4, hipMemcpyHostToDevice);
    free(h);
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (blocks > 4096) blocks = 4096;
  if (kind == 0) hipLaunchKernelGGL(neighbour<0>, dim3(blocks), dim3(256), 0, s, ops, out, iters);
  if (kind == 1) hipLaunchKernelGGL(neighbour<1>, dim3(blocks), dim3(256), 0, s, ops, out, iters);
  if (kind == 2) hipLaunchKernelGGL(neighbour<2>, dim3(blocks), dim3(256), 0, s, ops, out, iters);
  if (kind == 3) hipLaunchKernelGGL(neighbour<3>, dim3(blocks), dim3(256), 0, s, ops, out, iters);
  return (int)hipGetLastError();
}

#ifndef HAZARD_LIB
int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 200;
  hipStream_t sa, sb;
  hipStreamCreate(&sa);
  hipStreamCreate(&sb);
  unsigned* bad;
  float *ops, *out;
  hipMalloc(&bad, 4);
  hipMalloc(&ops, 65536 * 4);
  hipMalloc(&out, 1024 * 256 * 4);
  float* h = (float*)malloc(65536 * 4);
  srand(3);
  for (int i = 0; i < 65536; ++i) h[i] = (rand() / (float)RAND_MAX * 2.f - 1.f) * 0.25f;
  hipMemcpy(ops, h, 65536 * 4, hipMemcpyHostToDevice);
  const char* names[5] = {"v_mfma_f32_16x16x32_bf16", "v_mfma_f32_32x32x16_bf16", "v_mfma_f32_16x16x4_f32", "v_mfma_f32_16x16x32_f16", "none"};
  for (int use_lds = 0; use_lds < 2; ++use_lds)
    for (int kind = 0; kind < 5; ++kind) {
      hipMemset(bad, 0, 4);
      for (int r = 0; r < rounds; ++r) {
        // neighbour: 512 blocks x 4 waves, a few ms; victim: 1024 blocks x 4 waves, short, several per neighbour launch
        if (kind == 0) hipLaunchKernelGGL(neighbour<0>, dim3(512), dim3(256), 0, sb, ops, out, 20000);
        if (kind == 1) hipLaunchKernelGGL(neighbour<1>, dim3(512), dim3(256), 0, sb, ops, out, 10000);
        if (kind == 2) hipLaunchKernelGGL(neighbour<2>, dim3(512), dim3(256), 0, sb, ops, out, 20000);
        if (kind == 3) hipLaunchKernelGGL(neighbour<3>, dim3(512), dim3(256), 0, sb, ops, out, 20000);
        for (int v = 0; v < 8; ++v) hipLaunchKernelGGL(victim, dim3(1024), dim3(256), 0, sa, bad, (float*)nullptr, 2000, use_lds);
      }
      hipDeviceSynchronize();
      unsigned nb = 0;
      hipMemcpy(&nb, bad, 4, hipMemcpyDeviceToHost);
      printf("victim v_pk_fma_f32 vs 2 x v_fma_f32 (%s), neighbour %-26s: %u lanes with a mismatch, of %.3g lane-launches x 2000 iterations  (last error %d)\n",
             use_lds ? "with LDS round trip" : "registers only", names[kind], nb, (double)rounds * 8 * 1024 * 256, (int)hipGetLastError());
    }
  return 0;
}
#endif
