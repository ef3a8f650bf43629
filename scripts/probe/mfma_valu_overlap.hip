// Probe: do the matrix pipe and the VALU of ONE SIMD overlap (a) across two co-resident waves, (b) inside one wave's stream?
// 512-thread workgroups, one per CU: waves 0-3 (one per SIMD) run NM MFMAs (32x32x16 f16, 4 independent accumulators),
// waves 4-7 run NV VALU instructions (v_fma_f32 or v_sin_f32, 8 independent chains).  mode 0: MFMA waves only; 1: VALU waves
// only; 2: both; 3: one wave per SIMD carrying both, K VALU after every MFMA.      hipcc --offload-arch=gfx950 -O3 -o mvo ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int SIN>
__device__ __forceinline__ float valu_step(float x) {
  float y;
  if (SIN) asm volatile("v_sin_f32 %0, %1" : "=v"(y) : "v"(x));
  else asm volatile("v_fma_f32 %0, %1, %1, %1" : "=v"(y) : "v"(x));
  return y;
}

template <int SIN, int K>
__global__ __launch_bounds__(512) void probe(float* out, long long* cyc, int iters, int mode) {
  const int wave = threadIdx.x >> 6;
  const bool mf = (mode == 0 || mode == 2) ? wave < 4 : (mode == 3 ? wave < 4 : false);
  const bool va = (mode == 1 || mode == 2) ? wave >= 4 : false;
  f16v acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  h8 a, b;
  for (int r = 0; r < 8; ++r) { a[r] = (_Float16)(threadIdx.x * 0.001f + r); b[r] = (_Float16)(r * 0.5f); }
  float v[8];
  for (int r = 0; r < 8; ++r) v[r] = threadIdx.x * 0.01f + r;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  if (mode == 3) {
    if (wave < 4)
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
#pragma unroll
          for (int k = 0; k < K; ++k) v[(i * K + k) & 7] = valu_step<SIN>(v[(i * K + k) & 7]);
        }
      }
  } else if (mf) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    }
  } else if (va) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 4 * K; ++k) v[k & 7] = valu_step<SIN>(v[k & 7]);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int r = 0; r < 8; ++r) s += v[r];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

template <int SIN, int K>
void run(const char* name) {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 64);
  const int iters = 20000;
  printf("%s, K = %d VALU per MFMA (4 MFMA + %d VALU per iteration), %d iterations\n", name, K, 4 * K, iters);
  for (int mode = 0; mode < 4; ++mode) {
    hipMemset(cyc, 0, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<SIN, K><<<256, 512>>>(out, cyc, iters, mode);      // warm
    hipEventRecord(e0);
    probe<SIN, K><<<256, 512>>>(out, cyc, iters, mode);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    const char* mn[4] = {"MFMA waves only", "VALU waves only", "both, two waves per SIMD", "one wave per SIMD, interleaved"};
    printf("  mode %d %-32s %8.3f ms   cycles/iteration: wave0 %7.1f  wave4 %7.1f\n", mode, mn[mode], ms, (double)h[0] / iters, (double)h[4] / iters);
  }
}

int main() {
  run<0, 2>("v_fma_f32"); run<0, 4>("v_fma_f32"); run<0, 8>("v_fma_f32"); run<0, 16>("v_fma_f32");
  run<1, 1>("v_sin_f32"); run<1, 2>("v_sin_f32"); run<1, 4>("v_sin_f32");
  return 0;
}
