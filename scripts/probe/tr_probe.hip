// Probe of ds_read_b64_tr_b16 semantics on gfx950: LDS filled with u16 = element index; every lane passes
// the byte address lane*8 (+ optional variants) and prints the four 16-bit values it receives.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(unsigned* out, int mode) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int l = threadIdx.x;
  unsigned addr;
  if (mode == 0) addr = l * 8;                         // lane l -> elements 4l..4l+3
  else if (mode == 1) addr = (l & 15) * 8 + (l >> 4) * 1024;   // 16-lane groups on separate 512-element blocks
  else addr = (l & 15) * 128 + (l >> 4) * 8;           // lane i of a group -> row i of a [16][64] matrix, group g -> cols 4g..4g+3
  addr += (unsigned)(size_t)0;
  uint2 v;
  unsigned base = (unsigned)(size_t)lds;   // LDS offset of the array
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(base + addr) : "memory");
  out[l * 4 + 0] = v.x & 0xffff; out[l * 4 + 1] = v.x >> 16; out[l * 4 + 2] = v.y & 0xffff; out[l * 4 + 3] = v.y >> 16;
}
int main() {
  unsigned* d; hipMalloc(&d, 64 * 4 * 4);
  unsigned h[256];
  for (int mode = 0; mode < 3; ++mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) printf("  lane %2d: %4u %4u %4u %4u\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  }
  return 0;
}
