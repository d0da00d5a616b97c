// What does the 16-bit matrix pipe of THIS chip sustain?  (round 4: the ceiling the fp16x2 kernels are priced against.)
// MI355X_MICROARCH.md quotes 2.5 PFLOP/s dense at 2.4 GHz; under a real MFMA load the chip clocks to its power budget and the
// sustained figure depends on the operand DATA (zero operands toggle nothing).  This probe runs register-resident
// v_mfma_f32_32x32x16_f16 streams -- no LDS, no memory traffic: the matrix pipe alone -- for ~0.3 s per arm and prints
// executed TFLOP/s:   operands zero / random fp16,  1 / 2 waves per SIMD,  4 independent accumulators per wave.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_peak_probe tools/mfma_peak_probe.hip && tools/bin/mfma_peak_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void mfma_stream(const f16x8* __restrict__ ops, float* __restrict__ out, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  f16x8 a0 = ops[(t * 4 + 0) & 65535], a1 = ops[(t * 4 + 1) & 65535], b0 = ops[(t * 4 + 2) & 65535], b1 = ops[(t * 4 + 3) & 65535];
  f32x16 c0, c1, c2, c3;
  for (int e = 0; e < 16; ++e) { c0[e] = 0.f; c1[e] = 0.f; c2[e] = 0.f; c3[e] = 0.f; }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, c3, 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int e = 0; e < 16; ++e) s += c0[e] + c1[e] + c2[e] + c3[e];
  out[t] = s;
}

int main(int argc, char** argv) {
  // mfma_peak_probe [iters = 100000] [reps = 3]: bench.py runs a short form (30000, 2) for the `roofline.sustained` entries
  const int iters_arg = argc > 1 ? atoi(argv[1]) : 100000, reps = argc > 2 ? atoi(argv[2]) : 3;
  const int n_ops = 65536;
  std::vector<_Float16> h(n_ops * 8);
  f16x8* d_ops;
  float* d_out;
  hipMalloc(&d_ops, n_ops * sizeof(f16x8));
  hipMalloc(&d_out, 256 * 8 * 512 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int data = 0; data < 2; ++data) {
    srand(7);
    for (auto& v : h) v = data ? (_Float16)((rand() / (float)RAND_MAX * 2.f - 1.f) * 0.25f) : (_Float16)0.f;
    hipMemcpy(d_ops, h.data(), n_ops * sizeof(f16x8), hipMemcpyHostToDevice);
    for (int wps = 1; wps <= 2; ++wps) {
      const int threads = 256 * wps, blocks = 256 * 4;      // 4 blocks per CU resident? no: 1024 blocks queue; each CU runs them in turn
      const int iters = iters_arg;
      for (int rep = 0; rep < reps; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_stream, dim3(blocks), dim3(threads), 0, 0, d_ops, d_out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)blocks * (threads / 64) * iters * 16.0 * 2.0 * 32 * 32 * 16;
        printf("operands %-6s  %d wave(s)/SIMD-equivalent block of %3d threads  rep %d: %8.2f ms  %8.1f executed TFLOP/s\n",
               data ? "random" : "zero", wps, threads, rep, ms, flops / (ms * 1e-3) / 1e12);
      }
    }
  }
  return 0;
}
