// What bounds the head GEMM's epilogue (gemm_bf16x3_v3.hip)?  256 workgroups of 512 threads, each streaming "its tile":
// 256 rows x 256 bf16 columns of TWO planes (row pitch 1024 B, like the 512-wide head), with the access shapes an epilogue
// can use for a 16-byte-per-lane instruction:
//   pattern 0: 4 lanes per row piece  -> 16 rows x 64 B per instruction   (what the kernel does: half-sub-tile 16 x 32)
//   pattern 1: 8 lanes per row piece  ->  8 rows x 128 B per instruction  (one whole 128-byte line per 8 lanes)
//   pattern 2: 16 lanes per row piece ->  4 rows x 256 B per instruction
//   pattern 3: 64 lanes contiguous    ->  1 KiB contiguous per instruction
// op 0: stores only; 1: loads only (summed into a sink); 2: load one pair of planes, store another (heavy epilogue).
// Reports us per pass and bytes / clock / CU at the measured clock-agnostic rate (GB/s per CU).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int PAT, int OP>
__global__ __launch_bounds__(512) void k(const u32x4* __restrict__ in_hi, const u32x4* __restrict__ in_lo, u32x4* __restrict__ out_hi,
                                         u32x4* __restrict__ out_lo, unsigned* sink, int tiles_per_wg, int stagger) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
  constexpr int LPR = PAT == 0 ? 4 : PAT == 1 ? 8 : PAT == 2 ? 16 : 64;      // lanes per row piece
  constexpr int ROWS = 64 / LPR;                                               // rows per instruction
  constexpr int PIECE = LPR * 16;                                              // bytes per row piece
  constexpr int NCOLP = 256 / PIECE > 0 ? 256 / PIECE : 1;                     // pieces per wave row (wave: 64 rows x 128 cols = 256 B)
  if (stagger) { long long t0 = clock64(); while (clock64() - t0 < (long long)(blockIdx.x & 7) * stagger) __builtin_amdgcn_s_sleep(8); }
  u32x4 acc = {0, 0, 0, 0};
  for (int t = 0; t < tiles_per_wg; ++t) {
    const long long tile = (long long)blockIdx.x * tiles_per_wg + t;          // tile = 256 rows x 1024 B ... rows of pitch 1024 B: two tiles side by side
    const long long base = (tile >> 1) * 256 * 1024 + (tile & 1) * 512;       // bytes
    if (PAT < 3) {
      for (int rs = 0; rs < 64 / ROWS; ++rs)
        for (int cp = 0; cp < NCOLP; ++cp) {
          const int row = wm * 64 + rs * ROWS + lane / LPR;
          const long long off = (base + (long long)row * 1024 + wn * 256 + cp * PIECE + (lane % LPR) * 16) >> 4;
          if (OP >= 1) { u32x4 a = in_hi[off], b = in_lo[off]; acc += a; acc ^= b; }
          if (OP != 1) { u32x4 v = {(unsigned)off, acc.x, 2u, 3u}; out_hi[off] = v; out_lo[off] = v; }
        }
    } else {
      // contiguous: the wave's 64 x 256 B region is not contiguous in a pitch-1024 tensor: use a packed tile instead
      for (int i = 0; i < 16; ++i) {
        const long long off = ((tile * 256 * 256 * 2) >> 4) / 2 * 2 / 2 + (long long)(wave * 16 + i) * 64 + lane;   // packed 128 KiB per plane per tile
        if (OP >= 1) { u32x4 a = in_hi[off], b = in_lo[off]; acc += a; acc ^= b; }
        if (OP != 1) { u32x4 v = {(unsigned)off, acc.x, 2u, 3u}; out_hi[off] = v; out_lo[off] = v; }
      }
    }
  }
  if (OP >= 1 && acc.x == 0x12345678u) sink[0] = acc.y;
}
template <int PAT, int OP> static void run(const char* name, u32x4* a, u32x4* b, u32x4* c, u32x4* d, unsigned* sink, int stagger) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int tiles = 4, reps = 40;
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<PAT, OP>), dim3(256), dim3(512), 0, 0, a, b, c, d, sink, tiles, stagger);
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<PAT, OP>), dim3(256), dim3(512), 0, 0, a, b, c, d, sink, tiles, stagger);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps, bytes = 256.0 * tiles * 256 * 256 * 2 * 2 * (OP == 2 ? 2 : 1);
  printf("%-34s stagger %6d: %7.1f us / pass of %5.0f MB  -> %5.2f TB/s chip, %5.1f GB/s per CU, %5.1f us per tile\n", name, stagger, us, bytes / 1e6,
         bytes / us / 1e6, bytes / us / 1e3 / 256, us / tiles);
}
int main() {
  const size_t plane = 256ull * 4 * 256 * 256 * 2 + (1 << 20);
  u32x4 *a, *b, *c, *d; unsigned* sink;
  hipMalloc(&a, plane); hipMalloc(&b, plane); hipMalloc(&c, plane); hipMalloc(&d, plane); hipMalloc(&sink, 64);
  hipMemset(a, 1, plane); hipMemset(b, 2, plane);
  for (int st = 0; st <= 20000; st += 20000) {
    run<0, 0>("store 16 rows x 64 B", a, b, c, d, sink, st); run<1, 0>("store 8 rows x 128 B", a, b, c, d, sink, st);
    run<2, 0>("store 4 rows x 256 B", a, b, c, d, sink, st); run<3, 0>("store 1 KiB contiguous", a, b, c, d, sink, st);
    run<0, 1>("load 16 rows x 64 B", a, b, c, d, sink, st); run<1, 1>("load 8 rows x 128 B", a, b, c, d, sink, st);
    run<2, 1>("load 4 rows x 256 B", a, b, c, d, sink, st); run<3, 1>("load 1 KiB contiguous", a, b, c, d, sink, st);
    run<0, 2>("load+store 16 rows x 64 B", a, b, c, d, sink, st); run<1, 2>("load+store 8 rows x 128 B", a, b, c, d, sink, st);
    run<2, 2>("load+store 4 rows x 256 B", a, b, c, d, sink, st); run<3, 2>("load+store 1 KiB contiguous", a, b, c, d, sink, st);
  }
  return 0;
}
