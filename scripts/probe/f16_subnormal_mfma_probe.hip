// probe: does v_mfma_f32_32x32x16_f16 keep fp16 subnormal inputs? (A row of subnormals x B ones)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out, unsigned short abits) {
  h8 a, b;
  _Float16 av = __builtin_bit_cast(_Float16, abits);
  for (int j = 0; j < 8; ++j) { a[j] = av; b[j] = (_Float16)1.0f; }
  f32x16 c = {};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
  float* d; hipMalloc(&d, 4);
  unsigned short vals[] = {0x0001, 0x0200, 0x03ff, 0x0400, 0x3c00};
  for (unsigned short v : vals) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, v);
    float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    _Float16 f = __builtin_bit_cast(_Float16, v);
    printf("a=0x%04x (%.9g): sum of 16 products = %.9g (expected %.9g)\n", v, (double)(float)f, h, 16.0 * (double)(float)f);
  }
  return 0;
}
