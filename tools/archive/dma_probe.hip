// Probe of buffer_load_dwordx4 ... lds on gfx950: destination order, out-of-range behaviour, soffset vs range check.
// build: hipcc --offload-arch=gfx950 -O2 tools/dma_probe.hip -o /tmp/dma_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* x, float* y, unsigned nbytes, unsigned bad_off, unsigned soff, int mode) {
  __shared__ __attribute__((aligned(16))) float lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = -7.f;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, nbytes, 0x00020000);
  unsigned voff = threadIdx.x * 16;
  if (mode == 1 && (threadIdx.x & 1)) voff = bad_off;          // explicit out-of-range lanes
  if (mode == 2) voff = (63 - threadIdx.x) * 16;               // reversed source order
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + 256), 16, voff, soff, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) y[i] = lds[i];
}
int main() {
  const int n = 1024;
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = (float)i;
  float *x, *y;
  hipMalloc(&x, n * 4); hipMalloc(&y, 512 * 4);
  hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice);
  std::vector<float> o(512);
  auto run = [&](const char* name, unsigned nbytes, unsigned bad, unsigned soff, int mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, x, y, nbytes, bad, soff, mode);
    hipMemcpy(o.data(), y, 512 * 4, hipMemcpyDeviceToHost);
    printf("%s: lds[252..259] =", name);
    for (int i = 252; i < 260; ++i) printf(" %g", o[i]);
    printf(" | lane1 dst lds[260..263] = %g %g %g %g | lane63 dst lds[508..511] = %g %g %g %g\n", o[260], o[261], o[262], o[263], o[508], o[509], o[510], o[511]);
  };
  run("linear            ", n * 4, 0, 0, 0);
  run("odd lanes OOB     ", n * 4, 0xFFFFFFF0u, 0, 1);
  run("reversed source   ", n * 4, 0, 0, 2);
  run("soffset=1024B     ", n * 4, 0, 1024, 0);
  // range check vs soffset: num_records covers only 2048 B; lanes 0..63 voff 0..1008 (<2048) but voff+soff up to 3056
  run("nrec=2048,soff=2048", 2048, 0, 2048, 0);
  run("nrec=2048,soff=1040", 2048, 0, 1040, 0);
  return 0;
}
