#include <hip/hip_runtime.h>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
  __shared__ __attribute__((aligned(16))) __bf16 t[16 * 128];
  for (int i = threadIdx.x; i < 16 * 128; i += 64) t[i] = (__bf16)(float)i;
  __syncthreads();
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  // block rows 4*(g>>1)..+3? simple test: group g reads rows 4g..4g+3, cols 0..15
  __attribute__((address_space(3))) s16x4* p = (__attribute__((address_space(3))) s16x4*)(t + (4 * g + (i >> 2)) * 128 + (i & 3) * 4);
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = __builtin_bit_cast(float, (unsigned)(unsigned short)v[j] << 16);
}
int main() {
  float* d; hipMalloc(&d, 64 * 4 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l += 5) printf("lane %d: %g %g %g %g\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  return 0;
}
