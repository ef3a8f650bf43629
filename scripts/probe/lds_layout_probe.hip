// LDS access-pattern probe for the fused SIREN backward: counts SQ_LDS_BANK_CONFLICT for candidate image
// layouts under (a) forward ds_read_b64 fragments, (b) ds_read_b64_tr_b16 fragments, (c) ds_write_b64 staging.
// build: hipcc --offload-arch=gfx950 -O3 lds_layout_probe.hip -o lds_probe ; run under rocprofv3 --pmc
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short short4v __attribute__((ext_vector_type(4)));
typedef unsigned char uchar;

__device__ __forceinline__ uint2 lds_tr(const uchar* p) {
  short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)p);
  return __builtin_bit_cast(uint2, v);
}
__device__ __forceinline__ int swz5(int r) { return ((r & 3) << 3) | ((r >> 2) & 7); }

// layout 0: [row][128 cols] 256-B rows, 8-B units XOR swz5(row)            (current)
// layout 1: [row][128 cols] 256-B rows, no swizzle
// layout 2: [col/32][row][32 cols] 64-B rows, units XOR (row>>2)&7          (weights candidate)
// layout 3: [col/32][row][32 cols] 64-B rows, units XOR (row>>1)&7          (staging candidate)
// layout 4: [col/32][row][32 cols] 64-B rows, no swizzle
template <int L> __device__ __forceinline__ int addr(int row, int unit /* 8-byte unit = 4 cols, 0..31 */, int rows) {
  if (L == 0) return row * 256 + ((unit ^ swz5(row)) << 3);
  if (L == 1) return row * 256 + (unit << 3);
  const int blk = unit >> 3, ul = unit & 7;
  if (L == 2) return blk * rows * 64 + row * 64 + ((ul ^ ((row >> 2) & 7)) << 3);
  if (L == 3) return blk * rows * 64 + row * 64 + ((ul ^ ((row >> 1) & 7)) << 3);
  return blk * rows * 64 + row * 64 + (ul << 3);
}

template <int L, int MODE>
__global__ __launch_bounds__(256) void probe(unsigned* out, int iters) {
  extern __shared__ __attribute__((aligned(1024))) uchar sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, hf = lane >> 5, s16 = lane & 15, mhalf = (lane >> 4) & 1;
  for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<unsigned*>(sm)[i] = i;
  __syncthreads();
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {   // forward fragments: lane row = 32m + l31, units 8q+4t+hf and +2, 128 rows
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          uint2 a = *reinterpret_cast<const uint2*>(sm + addr<L>(32 * m + l31, 4 * s + hf, 128));
          uint2 b = *reinterpret_cast<const uint2*>(sm + addr<L>(32 * m + l31, 4 * s + hf + 2, 128));
          acc += a.x ^ a.y ^ b.x ^ b.y;
        }
    } else if (MODE == 1) {   // transposed fragments (register-chain k order): rows 16ks+4hf+(s16>>2) [+8], 128 rows
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int r0 = 16 * ks + 4 * hf + (s16 >> 2), u = 8 * m + 4 * mhalf + (s16 & 3);
          uint2 a = lds_tr(sm + addr<L>(r0, u, 128));
          uint2 b = lds_tr(sm + addr<L>(r0 + 8, u, 128));
          acc += a.x ^ a.y ^ b.x ^ b.y;
        }
    } else if (MODE == 2) {   // staging fragments (natural k order): rows 16ks+8hf+(s16>>2) [+4], 32-row image
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int r0 = 16 * ks + 8 * hf + (s16 >> 2), u = 8 * m + 4 * mhalf + (s16 & 3);
          uint2 a = lds_tr(sm + addr<L>(r0, u, 32));
          uint2 b = lds_tr(sm + addr<L>(r0 + 4, u, 32));
          acc += a.x ^ a.y ^ b.x ^ b.y;
        }
    } else {   // staging writes: lane = point row l31 (+32*wave), units 8q+2g+hf, 128-row image region per wave offset
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<uint2*>(sm + addr<L>(l31, 8 * q + 2 * g + hf, 32) + wave * 8192) = make_uint2(acc + q, it + g);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc + reinterpret_cast<unsigned*>(sm)[threadIdx.x];
}

template <int L, int MODE> void run(unsigned* out, const char* name) {
  hipFuncSetAttribute((const void*)probe<L, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<L, MODE>), dim3(256), dim3(256), 65536, 0, out, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<L, MODE>), dim3(256), dim3(256), 65536, 0, out, 2000);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s layout %d: %.3f ms\n", name, L, ms);
}

int main() {
  unsigned* out; hipMalloc(&out, 256 * 256 * 4);
  run<0, 0>(out, "fwd ds_read_b64"); run<1, 0>(out, "fwd ds_read_b64"); run<2, 0>(out, "fwd ds_read_b64"); run<4, 0>(out, "fwd ds_read_b64");
  run<0, 1>(out, "tr (chain k order)"); run<1, 1>(out, "tr (chain k order)"); run<2, 1>(out, "tr (chain k order)"); run<4, 1>(out, "tr (chain k order)");
  run<0, 2>(out, "tr (staging)"); run<1, 2>(out, "tr (staging)"); run<3, 2>(out, "tr (staging)"); run<4, 2>(out, "tr (staging)");
  run<0, 3>(out, "ds_write_b64 staging"); run<1, 3>(out, "ds_write_b64 staging"); run<3, 3>(out, "ds_write_b64 staging"); run<4, 3>(out, "ds_write_b64 staging");
  hipDeviceSynchronize();
  return 0;
}
