// Probe: HBM write bandwidth of 128-B pieces at a 256-B pitch (one cloud's 32-channel half of the [B,H,W,64] canvas)
// against contiguous writes, grid-stride against block-contiguous order, plain against nontemporal stores.
// build: hipcc --offload-arch=gfx950 -O2 tools/half_line_probe.hip -o tools/bin/half_line_probe   (run on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ void st(float* p, f4 v) {
  if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(p));
  else *reinterpret_cast<f4*>(p) = v;
}
// mode 0: grid-stride contiguous; 1: grid-stride, first 128 B of every 256 B
// mode 2: block-contiguous (block b owns `per` consecutive 16-B slots), full lines; 3: block-contiguous, half lines
template <bool NT>
__global__ void k(float* p, long cells, int mode, long per) {
  const f4 z = {1.f, 2.f, 3.f, 4.f};
  if (mode < 2) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x, nthr = (long)gridDim.x * blockDim.x;
    if (mode == 0) for (long i = t; i < cells * 16; i += nthr) st<NT>(p + i * 4, z);
    else for (long i = t; i < cells * 8; i += nthr) st<NT>(p + (i >> 3) * 64 + (i & 7) * 4, z);
  } else {
    const long b0 = (long)blockIdx.x * per;
    if (mode == 2) for (long i = threadIdx.x; i < per; i += blockDim.x) st<NT>(p + (b0 + i) * 4, z);
    else for (long i = threadIdx.x; i < per; i += blockDim.x) { const long j = b0 + i; st<NT>(p + (j >> 3) * 64 + (j & 7) * 4, z); }
  }
}
int main() {
  const long cells = 16L * 512 * 512;   // one B=16 canvas pair: 1.07 GB
  float* p;
  (void)hipMalloc(&p, cells * 256);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int nt = 0; nt < 2; ++nt)
    for (int mode = 0; mode < 4; ++mode)
      for (int blocks : {2048, 8192, 32768}) {
        const long slots = (mode & 1) ? cells * 8 : cells * 16;
        const long per = slots / blocks;
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
          (void)hipEventRecord(e0);
          if (nt) hipLaunchKernelGGL(k<true>, dim3(blocks), dim3(256), 0, 0, p, cells, mode, per);
          else hipLaunchKernelGGL(k<false>, dim3(blocks), dim3(256), 0, 0, p, cells, mode, per);
          (void)hipEventRecord(e1);
          (void)hipEventSynchronize(e1);
          (void)hipEventElapsedTime(&ms, e0, e1);
        }
        printf("nt %d mode %d blocks %5d: %.3f ms  %.2f TB/s\n", nt, mode, blocks, ms, slots * 16.0 / ms / 1e9);
      }
  return 0;
}
