// Issue-overlap probe (round 4): can ONE wave per SIMD hide its own VALU / transcendental / LDS instructions under its own
// MFMAs when they are interleaved in program order?  The fused SIREN backward (siren_bwd_x3.hip) runs one wave per SIMD
// with ~17 k MFMA cycles, ~18 k VALU cycles and ~14 k LDS cycles per round; round 3 concluded "a wave's own VALU does not
// issue under its MFMA's passes" from one whole-kernel experiment.  This measures it in isolation:
//   cycles per iteration (s_memtime, median over workgroups) of hand-ordered instruction streams, 256 threads = 1 wave / SIMD.
// build: hipcc --offload-arch=gfx950 -O3 issue_overlap_probe.hip -o issue_probe ; run: ./issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned char uchar;

#define SB __builtin_amdgcn_sched_barrier(0)
#define FMA(x) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(ka), "v"(kb))
#define SIN(x) asm volatile("v_sin_f32 %0, %0" : "+v"(x))
#define CVT(x, y) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(y))
#define LDSR(v, a, off) asm volatile("ds_read_b64 %0, %1 offset:" #off : "=v"(v) : "v"(a))
#define LDSW(a, v, off) asm volatile("ds_write_b64 %0, %1 offset:" #off : : "v"(a), "v"(v))

template <int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void probe(unsigned long long* out, float* sink, int iters) {
  extern __shared__ __attribute__((aligned(1024))) uchar sm[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) reinterpret_cast<unsigned*>(sm)[i] = i;
  __syncthreads();
  f32x16 acc[4];
  for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.0f + lane * 0.001f); b[i] = (__bf16)(0.5f); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = lane * 0.01f + i;
  const float ka = 0.999f, kb = 0.001f;
  const unsigned la = (unsigned)(uintptr_t)((__attribute__((address_space(3))) uchar*)sm) + lane * 8;
  uint2 lv[8];
  for (int i = 0; i < 8; ++i) lv[i] = make_uint2(0, 0);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (MODE == 0 || MODE == 2 || MODE == 4 || MODE == 5 || MODE == 6 || MODE == 8 || MODE == 9 || MODE == 10 || MODE == 11) {
        SB; acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k & 3], 0, 0, 0); SB;
      }
      if (MODE == 1 || MODE == 2) { FMA(v[0]); FMA(v[1]); FMA(v[2]); FMA(v[3]); }                       // 4 VALU per slot
      if (MODE == 3) { SIN(v[k & 7]); }
      if (MODE == 4) { SIN(v[k & 7]); FMA(v[(k + 1) & 7]); FMA(v[(k + 2) & 7]); FMA(v[(k + 3) & 7]); }   // 1 trans + 3 VALU
      if (MODE == 6) { FMA(v[0]); FMA(v[1]); FMA(v[2]); FMA(v[3]); FMA(v[4]); FMA(v[5]); FMA(v[6]); }    // 7 VALU per slot
      if (MODE == 9 || MODE == 12) { FMA(v[0]); FMA(v[1]); FMA(v[2]); FMA(v[3]); FMA(v[4]); FMA(v[5]); FMA(v[6]); FMA(v[7]); }   // 8 VALU
      if (MODE == 7 || MODE == 8) { LDSR(lv[(2 * k) & 7], la, 0); LDSR(lv[(2 * k + 1) & 7], la, 512); }  // 2 ds_read_b64 per slot
      if (MODE == 10) { LDSW(la, lv[0], 1024); LDSW(la, lv[1], 2048); }                                   // 2 ds_write_b64 per slot
      if (MODE == 11) { CVT(v[0], v[1]); CVT(v[2], v[3]); SIN(v[4]); SIN(v[5]); FMA(v[6]); FMA(v[7]); }  // mix: 2 cvt_pk, 2 trans, 2 fma
      if (MODE == 13) { CVT(v[0], v[1]); CVT(v[2], v[3]); SIN(v[4]); SIN(v[5]); FMA(v[6]); FMA(v[7]); }
    }
    if (MODE == 5) {                                            // 16 MFMAs first, then the 64 VALU as one block
#pragma unroll
      for (int k = 0; k < 16; ++k) { FMA(v[0]); FMA(v[1]); FMA(v[2]); FMA(v[3]); }
    }
    if (MODE == 7 || MODE == 8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  SB;
  float s = 0.f;
  for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
  for (int i = 0; i < 8; ++i) s += v[i] + __uint_as_float(lv[i].x) + __uint_as_float(lv[i].y);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x * WAVES + (threadIdx.x >> 6)] = t1 - t0;
  if (s == 12345.678f) sink[0] = s;
}

template <int MODE, int WAVES>
static double run(const char* what, int iters, unsigned long long* dout, float* sink, double base_mfma) {
  const int blocks = 256;
  hipLaunchKernelGGL((probe<MODE, WAVES>), dim3(blocks), dim3(64 * WAVES), 65536, 0, dout, sink, iters);
  hipLaunchKernelGGL((probe<MODE, WAVES>), dim3(blocks), dim3(64 * WAVES), 65536, 0, dout, sink, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * WAVES);
  hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double med = (double)h[h.size() / 2] / iters;
  printf("mode %2d waves/SIMD %d  %-72s %8.1f ticks/iter  (min %.1f max %.1f)\n", MODE, WAVES / 4, what, med, (double)h.front() / iters,
         (double)h.back() / iters);
  return med;
}

int main() {
  unsigned long long* dout; float* sink;
  hipMalloc(&dout, 256 * 8 * 8); hipMalloc(&sink, 4);
  const int it = 2000;
  hipFuncSetAttribute((const void*)probe<0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  printf("per iteration = 16 slots; s_memtime ticks (100 MHz constant clock on gfx9: compare RATIOS)\n");
  run<0, 4>("16 MFMA 32x32x16 bf16 (4 accumulators, round robin)", it, dout, sink, 0);
  run<1, 4>("64 v_fma (4 per slot), no MFMA", it, dout, sink, 0);
  run<2, 4>("16 x (MFMA, 4 v_fma)", it, dout, sink, 0);
  run<5, 4>("16 MFMA, then 64 v_fma as one block", it, dout, sink, 0);
  run<6, 4>("16 x (MFMA, 7 v_fma)", it, dout, sink, 0);
  run<12, 4>("128 v_fma (8 per slot), no MFMA", it, dout, sink, 0);
  run<9, 4>("16 x (MFMA, 8 v_fma)", it, dout, sink, 0);
  run<3, 4>("16 v_sin, no MFMA", it, dout, sink, 0);
  run<4, 4>("16 x (MFMA, v_sin, 3 v_fma)", it, dout, sink, 0);
  run<13, 4>("16 x (2 cvt_pk_bf16, 2 v_sin, 2 v_fma), no MFMA", it, dout, sink, 0);
  run<11, 4>("16 x (MFMA, 2 cvt_pk_bf16, 2 v_sin, 2 v_fma)", it, dout, sink, 0);
  run<7, 4>("32 ds_read_b64 (2 per slot), wait per iteration, no MFMA", it, dout, sink, 0);
  run<8, 4>("16 x (MFMA, 2 ds_read_b64), wait per iteration", it, dout, sink, 0);
  run<10, 4>("16 x (MFMA, 2 ds_write_b64)", it, dout, sink, 0);
  // two waves per SIMD: does the second wave's VALU hide under the first's MFMA?
  run<0, 8>("16 MFMA per wave, two waves per SIMD", it, dout, sink, 0);
  run<2, 8>("16 x (MFMA, 4 v_fma) per wave, two waves per SIMD", it, dout, sink, 0);
  run<9, 8>("16 x (MFMA, 8 v_fma) per wave, two waves per SIMD", it, dout, sink, 0);
  run<11, 8>("16 x (MFMA, 2 cvt, 2 sin, 2 fma) per wave, two waves per SIMD", it, dout, sink, 0);
  return 0;
}
