// Minimal reproducer (ROCm 7.2.0, AMD clang 22.0.0git roc-7.2.0, --offload-arch=gfx950 -O3; found in round 6 through
// tests/test_gpu_kernels.py::test_sparse_in_wgrad_vs_float64): __builtin_bit_cast applied DIRECTLY to an element of an ext_vector_type
// value reads element 0 whatever the index -- wrong() below stores (w[0], w[0]); right() copies the elements to scalars first and stores
// (w[0], w[1]).  Deterministic, visible in the ISA without a GPU:
//     hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only -o - tools/repro/bitcast_vector_element.hip | grep -A12 "^_Z5wrong"
//       global_load_dword v0, ...        <- ONE dword of the 8-byte element is fetched
//       v_mov_b32_e32 v1, v0             <- ... and written twice
// Run on a GPU: prints "wrong: 1 1   right: 1 2".  The library never applies __builtin_bit_cast to a vector element (grep of csrc/: every
// use is on a whole object or on an rvalue expression such as w[0] << 16); sparse_in_wgrad_kernel did for one session and summed the
// wrong canvas channel.  This is a front-end defect with a one-line workaround, NOT the intermittent interference DESIGN.md section 6
// describes (that one needs a concurrent neighbour and is not visible in the ISA).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__global__ void wrong(const u32x2* src, float* dst) {
  const u32x2 w = src[threadIdx.x];
  dst[2 * threadIdx.x] = __builtin_bit_cast(float, w[0]);
  dst[2 * threadIdx.x + 1] = __builtin_bit_cast(float, w[1]);
}
__global__ void right(const u32x2* src, float* dst) {
  const u32x2 w = src[threadIdx.x];
  const unsigned w0 = w[0], w1 = w[1];
  dst[2 * threadIdx.x] = __builtin_bit_cast(float, w0);
  dst[2 * threadIdx.x + 1] = __builtin_bit_cast(float, w1);
}

int main() {
  float h[2] = {1.f, 2.f}, o[4] = {0, 0, 0, 0};
  float *s, *d;
  (void)hipMalloc(&s, sizeof(h));
  (void)hipMalloc(&d, sizeof(o));
  (void)hipMemcpy(s, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(wrong, dim3(1), dim3(1), 0, 0, reinterpret_cast<const u32x2*>(s), d);
  hipLaunchKernelGGL(right, dim3(1), dim3(1), 0, 0, reinterpret_cast<const u32x2*>(s), d + 2);
  (void)hipMemcpy(o, d, sizeof(o), hipMemcpyDeviceToHost);
  printf("wrong: %g %g   right: %g %g\n", o[0], o[1], o[2], o[3]);
  return 0;
}
