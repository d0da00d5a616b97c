// Canary kernel (round 4): does some OTHER kernel running on the chip write into this workgroup's LDS or this wave's registers?
// Every workgroup fills its LDS with a pattern and every lane 48 registers, then re-checks both `rounds` times while the wave
// stays resident; mismatches are counted (LDS words / register values that changed without this kernel writing them).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/bin/libcanary.so tools/canary.hip      (driver: tools/canary_probe.py)
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(256) void canary_kernel(unsigned* __restrict__ bad, int lds_words, int rounds, unsigned* __restrict__ detail) {
  extern __shared__ unsigned lds[];
  const unsigned seed = blockIdx.x * 0x9E3779B9u + 0x1234567u;
  for (int i = threadIdx.x; i < lds_words; i += 256) lds[i] = seed ^ (i * 0x85EBCA6Bu);
  unsigned r[48];
#pragma unroll
  for (int k = 0; k < 48; ++k) r[k] = seed + threadIdx.x * 131u + k * 0x01000193u;
  __syncthreads();
  unsigned nl = 0, nv = 0;
  for (int it = 0; it < rounds; ++it) {
#pragma unroll
    for (int k = 0; k < 48; ++k) asm volatile("" : "+v"(r[k]));      // keeps all 48 live in VGPRs across the loop
    for (int i = threadIdx.x; i < lds_words; i += 256) {
      const unsigned v = lds[i], want = seed ^ (i * 0x85EBCA6Bu);
      if (v != want) {
        ++nl;
        if (detail) { detail[0] = i; detail[1] = v; detail[2] = want; detail[3] = blockIdx.x; }
        lds[i] = want;
      }
    }
    __builtin_amdgcn_s_sleep(8);
  }
#pragma unroll
  for (int k = 0; k < 48; ++k) nv += r[k] != seed + threadIdx.x * 131u + k * 0x01000193u;
  if (nl) atomicAdd(bad, nl);
  if (nv) atomicAdd(bad + 1, nv);
}

extern "C" int canary_launch(void* stream, unsigned* bad, int lds_bytes, int rounds, int blocks, unsigned* detail) {
  static int max_set = 0;
  if (lds_bytes > max_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(canary_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    max_set = lds_bytes;
  }
  hipLaunchKernelGGL(canary_kernel, dim3(blocks), dim3(256), lds_bytes, reinterpret_cast<hipStream_t>(stream), bad, lds_bytes / 4, rounds, detail);
  return (int)hipGetLastError();
}
