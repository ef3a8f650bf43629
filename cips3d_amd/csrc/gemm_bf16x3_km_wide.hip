// gemm_bf16x3_km_wide.hip — 256x256-tile, grouped form of the K-major split-bf16 GEMM
//   C[m][n] = sum_k A[k][m] * B[k][n],  A, B: row-major bf16 hi/lo planes with the contraction index as the row
// (the weight-gradient GEMMs of the CIPS head, dWb = X^T G with K = pixels; see gemm_bf16x3.hip for the 256x128
// kernel, the numerics and the ds_read_b64_tr_b16 fragment read this file reuses).
//
// At the head's shapes (M = N = 512, K = 4096 per image, 32 images) the 256x128 kernel is one tile per CU with
// every A element pulled through L2 four times and every B element twice (1.6 GB per GEMM for 536 MB of
// operands) and all eight LDS-DMA pieces of a k-tile issued in one burst.  Here
//   * the tile is 256 x 256 (8 waves as 4 x 2, 64 x 128 each): 1.07 GB through L2;
//   * one launch takes up to four problems of identical shape (a GROUP): a single 512 x 512 x 32-image problem is
//     only 128 such tiles, half a chip — the head's backward therefore issues the two weight-gradient GEMMs of a
//     block (dWb2 = a1^T g, dWb1 = x^T g1) together, 256 tiles.  (Splitting K instead and meeting in C through
//     fp32 atomics was measured at 2x the time of the whole GEMM: 16.7 M scalar atomics.)
//   * two 64 KiB LDS stages, the eight pieces of k-tile t+1 issued one per three MFMAs under k-tile t.
#include "common.h"
#include "../../include/cips3d_hip.h"
#include <stdlib.h>
#include <utility>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short short4v __attribute__((ext_vector_type(4)));
typedef short short8v __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

constexpr int BM = 256, BN = 256, BK = 32;
constexpr int ROW = 512;                         // bytes per k-row of an operand image (256 bf16)
constexpr int OFF_AHI = 0, OFF_ALO = 32 * ROW, OFF_BHI = 2 * 32 * ROW, OFF_BLO = 3 * 32 * ROW;
constexpr int STAGE = 4 * 32 * ROW;              // 65536
constexpr int NSTAGE = 2;
constexpr int PF = 36;                           // epilogue scratch pitch (fp32 [32][36] per wave, aliases stage 1)
constexpr int SMEM_BYTES = NSTAGE * STAGE;

constexpr int MAXG = 4;
// Convolution weight gradient (CONV instantiation): the contraction runs over ALL output pixels q of the batch; the A
// operand is dy as NHWC planes (row q, O columns), the B operand's row for (q, tap) is the input pixel
// (oy*stride - pad + ky, ox*stride - pad + kx) of the NHWC planes of x, or the zero row behind the last image.  A
// "group" is a tap, a "batch" entry a chunk of the pixel range (partial sums, added up by the caller).
struct KConv {
  int C, H, W, kw, stride, pad, Ho, Wo, ntap;
  long long zero_row;          // row index of the zero row
  int ktiles, nchunks;         // the pixel range is ktiles 32-row k-tiles; chunk c takes k-tiles [c*ktiles/nchunks, (c+1)*ktiles/nchunks)
};
struct KArgs {
  cips_gemm_x3_desc d;               // shape, leading dimensions, strides (common to the group)
  const void *A_hi[MAXG], *A_lo[MAXG], *B_hi[MAXG], *B_lo[MAXG];
  float* C[MAXG];
  int tiles_m, tiles_n, ngroups, total;
  KConv cv;
};

template <bool CONV>
__global__ __launch_bounds__(512) void gemm_bf16x3_km_wide_kernel(KArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const cips_gemm_x3_desc& d = g.d;
  const int tid = threadIdx.x;
  const int lane0 = tid & 63;
  const int uw = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = uw, wm = wave >> 1, wn = wave & 1;       // 4 x 2 waves, 64 x 128 outputs each
  const int M = d.M, N = d.N;
  const int nk_all = d.K / BK;
  const unsigned sbase = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)smem);

  for (int tseq = blockIdx.x; tseq < g.total; tseq += gridDim.x) {
    int lane = lane0;
    asm volatile("" : "+v"(lane));               // keep per-lane address math inside the persistent loop
    const int l31 = lane & 31, hf = lane >> 5, s16 = lane & 15, mhalf = (lane >> 4) & 1;
    int bid = tseq;
    {
      const int nx = 8;
      int q = g.total / nx, r = g.total % nx;
      int xcd = bid % nx, idx = bid / nx;
      int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
      bid = base + idx;
    }
    const int tn = bid % g.tiles_n;
    const int tm = (bid / g.tiles_n) % g.tiles_m;
    const int bz = (bid / (g.tiles_n * g.tiles_m)) % d.batch;
    const int gi = bid / (g.tiles_n * g.tiles_m * d.batch);       // problem of the group (uniform); CONV: the tap
    const int m0 = tm * BM, n0 = tn * BN;
    const int gp = CONV ? 0 : gi;
    // CONV: ragged chunks of the pixel range (any chunk count fills the chip: powers of two left 144 of 256 CUs busy)
    const long long kt0 = CONV ? (long long)bz * g.cv.ktiles / g.cv.nchunks : 0;
    const int nk = CONV ? (int)((long long)(bz + 1) * g.cv.ktiles / g.cv.nchunks - kt0) : nk_all;
    const u16* Ahi = (const u16*)g.A_hi[gp] + (CONV ? kt0 * BK * d.lda : (long long)bz * d.strideA);
    const u16* Alo = (const u16*)g.A_lo[gp] + (CONV ? kt0 * BK * d.lda : (long long)bz * d.strideA);
    const u16* Bhi = (const u16*)g.B_hi[gp] + (CONV ? 0 : (long long)bz * d.strideB);
    const u16* Blo = (const u16*)g.B_lo[gp] + (CONV ? 0 : (long long)bz * d.strideB);
    float* Cg = g.C[gp] + (CONV ? (long long)gi * d.M * d.ldc : 0);       // CONV: strideC spans all taps of a chunk
    const int tap_ky = CONV ? gi / g.cv.kw : 0, tap_kx = CONV ? gi - tap_ky * g.cv.kw : 0;

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // LDS-DMA: one piece = 2 k-rows x 512 B of one plane (lane -> row lane>>5, 16-byte chunk lane&31); 16 pieces
    // per plane and k-tile, 64 in all, 8 per wave.  The image stores 32-byte column pairs XOR-swizzled by
    // 2*(k&3) — in the source address here, in the fragment reads below.  k&3 = 2*(piece&1) + (lane>>5): two
    // lane offsets per operand, everything else is scalar.
    const int lh = lane >> 5, c16 = lane & 31;
    unsigned offA[2], offB[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      const int k3 = 2 * par + lh;
      const int pr = (c16 >> 1) ^ (2 * k3);
      int m = m0 + (pr * 2 + (c16 & 1)) * 8, n = n0 + (pr * 2 + (c16 & 1)) * 8;
      m = (m < M) ? m : 0;                       // clamped columns only feed outputs that are never stored
      n = (n < N) ? n : 0;
      offA[par] = (unsigned)(lh * d.lda + m) * 2u;
      offB[par] = (unsigned)(lh * d.ldb + n) * 2u;
    }
    // SGPR-base form through asm (see gemm_bf16x3_wide.hip): uniform row pointer + one 32-bit offset per lane
    auto dma = [&](const u16* p, unsigned off, unsigned char* lds_base) {
      const unsigned la = sbase + (unsigned)(lds_base - smem);
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(off), "s"(p), "s"(la) : "memory");
    };
    // CONV: byte offset of this lane's B row for contraction row q (a global output-pixel index) — its column part
    // comes from offB[] with the lane's k-row term removed (computed below with lh = 0 semantics)
    unsigned colB[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) colB[par] = offB[par] - (unsigned)(lh * d.ldb) * 2u;
    auto conv_row_off = [&](long long q, int par) -> unsigned {
      const int hw = g.cv.Ho * g.cv.Wo;
      const int b = (int)(q / hw), r = (int)(q - (long long)b * hw);
      const int oy = r / g.cv.Wo, ox = r - oy * g.cv.Wo;
      const int iy = oy * g.cv.stride - g.cv.pad + tap_ky, ix = ox * g.cv.stride - g.cv.pad + tap_kx;
      const bool ok = (unsigned)iy < (unsigned)g.cv.H && (unsigned)ix < (unsigned)g.cv.W;
      const long long row = ok ? ((long long)b * g.cv.H + iy) * g.cv.W + ix : g.cv.zero_row;
      return (unsigned)(row * d.ldb * 2) + colB[par];
    };
    unsigned cvoff[2] = {0, 0};            // offsets of the two B pieces (idx = uw, uw + 8) of the k-tile being issued
    auto conv_prep = [&](int k0) {
      if constexpr (CONV) {
        const long long qb = kt0 * BK + k0 + lh;
        cvoff[0] = conv_row_off(qb + 2 * uw, uw & 1);
        cvoff[1] = conv_row_off(qb + 2 * (uw + 8), uw & 1);      // (uw + 8) & 1 == uw & 1
      }
    };
    auto dma_piece = [&](int pc, int k0, unsigned char* s) {      // pc = 0..7: pieces uw + 8*(pc>>2) of plane pc&3
      const int idx = uw + 8 * (pc >> 2), which = pc & 3;
      const long long rowoff = (long long)(k0 + 2 * idx);
      if (which == 0) dma(Ahi + rowoff * d.lda, offA[idx & 1], s + OFF_AHI + idx * 1024);
      else if (which == 1) dma(Alo + rowoff * d.lda, offA[idx & 1], s + OFF_ALO + idx * 1024);
      else if constexpr (CONV) dma(which == 2 ? Bhi : Blo, cvoff[pc >> 2], s + (which == 2 ? OFF_BHI : OFF_BLO) + idx * 1024);
      else if (which == 2) dma(Bhi + rowoff * d.ldb, offB[idx & 1], s + OFF_BHI + idx * 1024);
      else dma(Blo + rowoff * d.ldb, offB[idx & 1], s + OFF_BLO + idx * 1024);
    };

    // transpose-read fragments: lane i = column col0 + (lane&31), k = 16ks + 8hf + {0..7}; the address is one of two
    // lane bases per operand (ks = 0 / 1 differ in the swizzle term) plus immediates
    auto fbase = [&](int colw, int ks) -> unsigned {
      const int kb = 16 * ks + 8 * hf + (s16 >> 2);
      const int col = colw + 16 * mhalf + 4 * (s16 & 3);
      const int pr = (col >> 4) ^ (2 * (kb & 3));
      return sbase + kb * ROW + pr * 32 + (col & 15) * 2;
    };
    // (kb & 3) = (s16 >> 2) & 3 for both ks; adding 32 columns flips bit 1 of col>>4: XOR with 2*(kb&3) keeps that a
    // plain +64 bytes only when the touched bits do not overlap — they do (bits 1..2), so each 32-column tile gets
    // its own base: 2 (A) + 4 (B) bases per ks.
    unsigned fa[2][2], fb[2][4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[ks][i] = fbase(wm * 64 + i * 32, ks);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[ks][j] = fbase(wn * 128 + j * 32, ks);
    }
#define LDS_TR(a) __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(uintptr_t)(a))
    auto frag = [&](unsigned base, int off) -> bf16x8 {
      short4v a = LDS_TR(base + off);
      short4v b = LDS_TR(base + off + 4 * ROW);
      short8v v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
      return __builtin_bit_cast(bf16x8, v);
    };
    auto compute = [&](int stage, bool issue_next, int next_stage, int k0n) {
      unsigned char* sn = smem + next_stage * STAGE;
      const int so = stage * STAGE;
      if (issue_next) conv_prep(k0n);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 ah[2], al[2], bh[4], bl[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) { ah[i] = frag(fa[ks][i], so + OFF_AHI); al[i] = frag(fa[ks][i], so + OFF_ALO); }
#pragma unroll
        for (int j = 0; j < 4; ++j) { bh[j] = frag(fb[ks][j], so + OFF_BHI); bl[j] = frag(fb[ks][j], so + OFF_BLO); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 24; ++m) {
          const int pass = m >> 3, i = (m >> 2) & 1, j = m & 3;
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pass == 0 ? al[i] : ah[i], pass == 1 ? bl[j] : bh[j], acc[i][j], 0, 0, 0);
          if (ks == 0 && (m % 3) == 2) {
            __builtin_amdgcn_sched_barrier(0);
            if (issue_next) dma_piece(m / 3, k0n, sn);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };

    if (nk > 0) {
      conv_prep(0);
#pragma unroll
      for (int pc = 0; pc < 8; ++pc) dma_piece(pc, 0, smem);
    }
    for (int kt = 0; kt < nk; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      compute(kt & 1, kt + 1 < nk, (kt + 1) & 1, (kt + 1) * BK);
    }
    __syncthreads();

    // ---- epilogue: 32 x 32 sub-tiles through a per-wave fp32 scratch, 16-byte row-contiguous stores / atomics
    float* sc_f = reinterpret_cast<float*>(smem + STAGE) + wave * (32 * PF);
    const long long cb = (long long)bz * d.strideC;
    const int h_rr = lane >> 2, h_c8 = (lane & 3) * 8;
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      const int si = st >> 2, jj = st & 3;
      const int row0 = m0 + wm * 64 + si * 32, col0 = n0 + wn * 128 + jj * 32;
#pragma unroll
      for (int r = 0; r < 16; ++r) sc_f[mfma_row(r, hf) * PF + l31] = (st < 4 ? acc[0][jj][r] : acc[1][jj][r]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int row = row0 + h_rr + 16 * c, col = col0 + h_c8;
        const float4 a = *reinterpret_cast<const float4*>(sc_f + (h_rr + 16 * c) * PF + h_c8);
        const float4 b = *reinterpret_cast<const float4*>(sc_f + (h_rr + 16 * c) * PF + h_c8 + 4);
        if (row < M && col < N) {
          float* q = Cg + cb + (long long)row * d.ldc + col;
          *reinterpret_cast<float4*>(q) = a;
          *reinterpret_cast<float4*>(q + 4) = b;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Round 3: the same tile, LDS image, DMA pieces and fragment reads under the schedule of gemm_bf16x3_v3.hip —
//   * the 12 operand fragments of a k-step live in TWO register sets: the transpose reads of k-step j+1 are issued
//     between the MFMAs of k-step j (one fragment = two ds_read_b64_tr_b16 per two MFMAs) instead of as a 24-read burst
//     in front of them (the two waves of a SIMD run in lockstep behind the barriers: neither covers the other's burst);
//   * ONE barrier per k-tile, four MFMAs into its second k-step: it publishes k-tile t+1 (DMA'd a whole k-tile period
//     earlier, not half of one) and frees the current stage for k-tile t+2, whose eight pieces follow one per two MFMAs.
// Same MFMA order per accumulator as the kernel above: results are bit-identical.
template <typename F, int... I>
__device__ __forceinline__ void kstatic_for(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
// fragment order of a k-step: 0 a_lo[0], 1..4 b_hi[0..3], 5 a_lo[1], 6 a_hi[0], 7..10 b_lo[0..3], 11 a_hi[1]
__device__ __forceinline__ constexpr bool kf_is_a(int q) { return q == 0 || q == 5 || q == 6 || q == 11; }
__device__ __forceinline__ constexpr int kf_idx(int q) { return q == 0 || q == 6 ? 0 : q == 5 || q == 11 ? 1 : q <= 4 ? q - 1 : q - 7; }
__device__ __forceinline__ constexpr int kf_plane(int q) { return q == 0 || q == 5 ? OFF_ALO : q == 6 || q == 11 ? OFF_AHI : q <= 4 ? OFF_BHI : OFF_BLO; }
__device__ __forceinline__ constexpr int km_a(int m) { return (m >> 3) == 0 ? (((m >> 2) & 1) ? 5 : 0) : (((m >> 2) & 1) ? 11 : 6); }
__device__ __forceinline__ constexpr int km_b(int m) { return (m >> 3) == 1 ? 7 + (m & 3) : 1 + (m & 3); }

template <bool CONV>
__global__ __launch_bounds__(512) void gemm_bf16x3_km_v3_kernel(KArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const cips_gemm_x3_desc& d = g.d;
  const int tid = threadIdx.x;
  const int lane0 = tid & 63;
  const int uw = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = uw, wm = wave >> 1, wn = wave & 1;       // 4 x 2 waves, 64 x 128 outputs each
  const int M = d.M, N = d.N;
  const int nk_all = d.K / BK;
  const unsigned sbase = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)smem);

  for (int tseq = blockIdx.x; tseq < g.total; tseq += gridDim.x) {
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int l31 = lane & 31, hf = lane >> 5, s16 = lane & 15, mhalf = (lane >> 4) & 1;
    int bid = tseq;
    {
      const int nx = 8;
      int q = g.total / nx, r = g.total % nx;
      int xcd = bid % nx, idx = bid / nx;
      int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
      bid = base + idx;
    }
    const int tn = bid % g.tiles_n;
    const int tm = (bid / g.tiles_n) % g.tiles_m;
    const int bz = (bid / (g.tiles_n * g.tiles_m)) % d.batch;
    const int gi = bid / (g.tiles_n * g.tiles_m * d.batch);
    const int m0 = tm * BM, n0 = tn * BN;
    const int gp = CONV ? 0 : gi;
    const long long kt0 = CONV ? (long long)bz * g.cv.ktiles / g.cv.nchunks : 0;
    const int nk = CONV ? (int)((long long)(bz + 1) * g.cv.ktiles / g.cv.nchunks - kt0) : nk_all;     // >= 2 (host)
    const u16* Ahi = (const u16*)g.A_hi[gp] + (CONV ? kt0 * BK * d.lda : (long long)bz * d.strideA);
    const u16* Alo = (const u16*)g.A_lo[gp] + (CONV ? kt0 * BK * d.lda : (long long)bz * d.strideA);
    const u16* Bhi = (const u16*)g.B_hi[gp] + (CONV ? 0 : (long long)bz * d.strideB);
    const u16* Blo = (const u16*)g.B_lo[gp] + (CONV ? 0 : (long long)bz * d.strideB);
    float* Cg = g.C[gp] + (CONV ? (long long)gi * d.M * d.ldc : 0);
    const int tap_ky = CONV ? gi / g.cv.kw : 0, tap_kx = CONV ? gi - tap_ky * g.cv.kw : 0;

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int lh = lane >> 5, c16 = lane & 31;
    unsigned offA[2], offB[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      const int k3 = 2 * par + lh;
      const int pr = (c16 >> 1) ^ (2 * k3);
      int m = m0 + (pr * 2 + (c16 & 1)) * 8, n = n0 + (pr * 2 + (c16 & 1)) * 8;
      m = (m < M) ? m : 0;
      n = (n < N) ? n : 0;
      offA[par] = (unsigned)(lh * d.lda + m) * 2u;
      offB[par] = (unsigned)(lh * d.ldb + n) * 2u;
    }
    auto dma = [&](const u16* p, unsigned off, unsigned la) {
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(off), "s"(p), "s"(la) : "memory");
    };
    unsigned colB[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) colB[par] = offB[par] - (unsigned)(lh * d.ldb) * 2u;
    auto conv_row_off = [&](long long q, int par) -> unsigned {
      const int hw = g.cv.Ho * g.cv.Wo;
      const int b = (int)(q / hw), r = (int)(q - (long long)b * hw);
      const int oy = r / g.cv.Wo, ox = r - oy * g.cv.Wo;
      const int iy = oy * g.cv.stride - g.cv.pad + tap_ky, ix = ox * g.cv.stride - g.cv.pad + tap_kx;
      const bool ok = (unsigned)iy < (unsigned)g.cv.H && (unsigned)ix < (unsigned)g.cv.W;
      const long long row = ok ? ((long long)b * g.cv.H + iy) * g.cv.W + ix : g.cv.zero_row;
      return (unsigned)(row * d.ldb * 2) + colB[par];
    };
    unsigned cvoff[2] = {0, 0};
    auto conv_prep = [&](int k0) {
      if constexpr (CONV) {
        const long long qb = kt0 * BK + k0 + lh;
        cvoff[0] = conv_row_off(qb + 2 * uw, uw & 1);
        cvoff[1] = conv_row_off(qb + 2 * (uw + 8), uw & 1);
      }
    };
    auto dma_piece = [&](int pc, int k0, unsigned st) {            // st: LDS byte offset of the stage
      const int idx = uw + 8 * (pc >> 2), which = pc & 3;
      const long long rowoff = (long long)(k0 + 2 * idx);
      const unsigned la = sbase + st + (unsigned)idx * 1024u;
      if (which == 0) dma(Ahi + rowoff * d.lda, offA[idx & 1], la + OFF_AHI);
      else if (which == 1) dma(Alo + rowoff * d.lda, offA[idx & 1], la + OFF_ALO);
      else if constexpr (CONV) dma(which == 2 ? Bhi : Blo, cvoff[pc >> 2], la + (which == 2 ? OFF_BHI : OFF_BLO));
      else if (which == 2) dma(Bhi + rowoff * d.ldb, offB[idx & 1], la + OFF_BHI);
      else dma(Blo + rowoff * d.ldb, offB[idx & 1], la + OFF_BLO);
    };
    auto fbase = [&](int colw, int ks) -> unsigned {
      const int kb = 16 * ks + 8 * hf + (s16 >> 2);
      const int col = colw + 16 * mhalf + 4 * (s16 & 3);
      const int pr = (col >> 4) ^ (2 * (kb & 3));
      return sbase + kb * ROW + pr * 32 + (col & 15) * 2;
    };
    unsigned fa[2][2], fb[2][4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[ks][i] = fbase(wm * 64 + i * 32, ks);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[ks][j] = fbase(wn * 128 + j * 32, ks);
    }
    auto frag = [&](unsigned base, int off) -> bf16x8 {
      short4v a = LDS_TR(base + off);
      short4v b = LDS_TR(base + off + 4 * ROW);
      short8v v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
      return __builtin_bit_cast(bf16x8, v);
    };
    bf16x8 F0[12], F1[12];
    auto rd = [&](auto Q_, const unsigned (&A)[2], const unsigned (&B)[4]) -> bf16x8 {
      constexpr int q = decltype(Q_)::value;
      if constexpr (kf_is_a(q)) return frag(A[kf_idx(q)], kf_plane(q));
      else return frag(B[kf_idx(q)], kf_plane(q));
    };

    // one k-tile.  MODE 0: steady state (DMA of k-tile kt+2, fragments of k-tile kt+1); 1: next-to-last (fragments only);
    // 2: last
    auto ktile = [&](auto MODE_, int kt) {
      constexpr int MODE = decltype(MODE_)::value;
      const unsigned cur = (unsigned)(kt & 1) * STAGE, nxt = STAGE - cur;
      unsigned a1[2] = {fa[1][0] + cur, fa[1][1] + cur}, b1[4] = {fb[1][0] + cur, fb[1][1] + cur, fb[1][2] + cur, fb[1][3] + cur};
      unsigned a0n[2] = {fa[0][0] + nxt, fa[0][1] + nxt}, b0n[4] = {fb[0][0] + nxt, fb[0][1] + nxt, fb[0][2] + nxt, fb[0][3] + nxt};
      asm volatile("" : "+v"(a1[0]), "+v"(a1[1]), "+v"(b1[0]), "+v"(b1[1]), "+v"(b1[2]), "+v"(b1[3]));
      asm volatile("" : "+v"(a0n[0]), "+v"(a0n[1]), "+v"(b0n[0]), "+v"(b0n[1]), "+v"(b0n[2]), "+v"(b0n[3]));
      if constexpr (MODE == 0) conv_prep((kt + 2) * BK);
      __builtin_amdgcn_sched_barrier(0);
      // k-step a: MFMAs on set 0, the twelve fragments of k-step b into set 1 (one per two MFMAs)
      kstatic_for(std::make_integer_sequence<int, 24>{}, [&](auto M_) {
        constexpr int m = decltype(M_)::value;
        acc[(m >> 2) & 1][m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F0[km_a(m)], F0[km_b(m)], acc[(m >> 2) & 1][m & 3], 0, 0, 0);
        if constexpr ((m & 1) == 0) {
          constexpr int q = m >> 1;
          F1[q] = rd(std::integral_constant<int, q>{}, a1, b1);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      // k-step b
      kstatic_for(std::make_integer_sequence<int, 24>{}, [&](auto M_) {
        constexpr int m = decltype(M_)::value;
        acc[(m >> 2) & 1][m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F1[km_a(m)], F1[km_b(m)], acc[(m >> 2) & 1][m & 3], 0, 0, 0);
        if constexpr (m == 3) {
          // every read of stage `cur` has returned, this wave's pieces of k-tile kt+1 have landed (issued a k-tile period
          // ago): behind the barrier stage `nxt` is readable and `cur` writable
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        if constexpr (MODE == 0) {
          if constexpr (m >= 4 && m <= 18 && (m & 1) == 0) dma_piece((m - 4) >> 1, (kt + 2) * BK, cur);
        }
        if constexpr (MODE <= 1) {
          if constexpr (m >= 5 && m <= 15 && (m & 1) == 1) {
            constexpr int q = m - 5;
            F0[q] = rd(std::integral_constant<int, q>{}, a0n, b0n);
            F0[q + 1] = rd(std::integral_constant<int, q + 1>{}, a0n, b0n);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    };

    // ---- tile start: k-tiles 0 and 1 requested; k-tile 0 has landed everywhere; its first fragments
    conv_prep(0);
#pragma unroll
    for (int pc = 0; pc < 8; ++pc) dma_piece(pc, 0, 0);
    conv_prep(BK);
#pragma unroll
    for (int pc = 0; pc < 8; ++pc) dma_piece(pc, BK, STAGE);
    asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
    kstatic_for(std::make_integer_sequence<int, 12>{}, [&](auto Q_) { F0[decltype(Q_)::value] = rd(Q_, fa[0], fb[0]); });
    __builtin_amdgcn_sched_barrier(0);
    for (int kt = 0; kt < nk - 2; ++kt) ktile(std::integral_constant<int, 0>{}, kt);
    ktile(std::integral_constant<int, 1>{}, nk - 2);
    ktile(std::integral_constant<int, 2>{}, nk - 1);
    __syncthreads();

    // ---- epilogue: 32 x 32 sub-tiles through a per-wave fp32 scratch (aliases stage 1), 16-byte row-contiguous stores
    float* sc_f = reinterpret_cast<float*>(smem + STAGE) + wave * (32 * PF);
    const long long cb = (long long)bz * d.strideC;
    const int h_rr = lane >> 2, h_c8 = (lane & 3) * 8;
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      const int si = st >> 2, jj = st & 3;
      const int row0 = m0 + wm * 64 + si * 32, col0 = n0 + wn * 128 + jj * 32;
#pragma unroll
      for (int r = 0; r < 16; ++r) sc_f[mfma_row(r, hf) * PF + l31] = (st < 4 ? acc[0][jj][r] : acc[1][jj][r]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int row = row0 + h_rr + 16 * c, col = col0 + h_c8;
        const float4 a = *reinterpret_cast<const float4*>(sc_f + (h_rr + 16 * c) * PF + h_c8);
        const float4 b = *reinterpret_cast<const float4*>(sc_f + (h_rr + 16 * c) * PF + h_c8 + 4);
        if (row < M && col < N) {
          float* q = Cg + cb + (long long)row * d.ldc + col;
          *reinterpret_cast<float4*>(q) = a;
          *reinterpret_cast<float4*>(q + 4) = b;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __syncthreads();
  }
}

}  // namespace

// Grouped entry: descs[0..ngroups) must agree in M, N, K, batch, leading dimensions and strides; only the operand
// and output pointers differ.  Returns hipErrorNotSupported when the shape does not qualify (caller falls back to
// one cips_gemm_bf16x3_km per problem).
extern "C" int cips_gemm_bf16x3_km_grouped(const cips_gemm_x3_desc* descs, int ngroups, cips_stream_t stream) {
  if (!descs || ngroups < 1) return (int)hipErrorInvalidValue;
  if (ngroups > MAXG) return (int)hipErrorNotSupported;
  const cips_gemm_x3_desc* d = &descs[0];
  if (d->M <= 0 || d->N <= 0 || d->K <= 0 || d->batch <= 0) return (int)hipErrorInvalidValue;
  if ((d->N & 7) || (d->M & 7) || (d->ldc & 3) || (d->strideC & 3) || (d->K & 31) || (d->lda & 7) || (d->ldb & 7) ||
      (d->strideA & 7) || (d->strideB & 7) || d->M < 256 || d->N < 256)
    return (int)hipErrorNotSupported;
  KArgs g;
  g.d = *d;
  for (int i = 0; i < ngroups; ++i) {
    const cips_gemm_x3_desc& e = descs[i];
    if (!e.C || !e.A_hi || !e.A_lo || !e.B_hi || !e.B_lo) return (int)hipErrorInvalidValue;
    if (e.M != d->M || e.N != d->N || e.K != d->K || e.batch != d->batch || e.lda != d->lda || e.ldb != d->ldb ||
        e.ldc != d->ldc || e.strideA != d->strideA || e.strideB != d->strideB || e.strideC != d->strideC)
      return (int)hipErrorNotSupported;
    if (e.P_hi || e.T_hi || e.mask || e.add || e.rgb_g || e.C_unmasked || e.mask_out || e.res_hi || e.act)
      return (int)hipErrorNotSupported;
    g.A_hi[i] = e.A_hi; g.A_lo[i] = e.A_lo; g.B_hi[i] = e.B_hi; g.B_lo[i] = e.B_lo; g.C[i] = e.C;
  }
  g.tiles_m = (d->M + BM - 1) / BM;
  g.tiles_n = (d->N + BN - 1) / BN;
  g.ngroups = ngroups;
  long long total = (long long)g.tiles_m * g.tiles_n * d->batch * ngroups;
  if (total > 0x7fffffffLL) return (int)hipErrorInvalidValue;
  g.total = (int)total;
  static int ncu = 0;
  CIPS_PER_DEVICE(ncu, 0);
  if (!ncu) {
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
    if (ncu <= 0) ncu = 256;
    ncu = (ncu / 8) * 8;
    (void)hipFuncSetAttribute((const void*)gemm_bf16x3_km_wide_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm_bf16x3_km_v3_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
  }
  const int grid = g.total < ncu ? g.total : ncu;
  if (d->K >= 2 * BK)                         // else: the one-k-tile form
    hipLaunchKernelGGL(gemm_bf16x3_km_v3_kernel<false>, dim3(grid), dim3(512), SMEM_BYTES, (hipStream_t)stream, g);
  else
    hipLaunchKernelGGL(gemm_bf16x3_km_wide_kernel<false>, dim3(grid), dim3(512), SMEM_BYTES, (hipStream_t)stream, g);
  return CIPS_CHECK_LAUNCH();
}

// Convolution weight gradient (see include/cips3d_hip.h): part[chunk][tap][o][c] = sum over the chunk's output pixels q
// of dy[q][o] * x[pixel(q) (+) tap][c]
extern "C" int cips_conv2d_x3_wgrad(const cips_conv_wgrad_desc* c, cips_stream_t stream) {
  if (!c || c->B <= 0 || c->C <= 0 || c->O <= 0 || c->H <= 0 || c->W <= 0 || c->kh <= 0 || c->kw <= 0 || c->stride <= 0 ||
      c->pad < 0 || c->nchunks <= 0 || !c->part)
    return (int)hipErrorInvalidValue;
  const int Ho = (c->H + 2 * c->pad - c->kh) / c->stride + 1, Wo = (c->W + 2 * c->pad - c->kw) / c->stride + 1;
  if (Ho <= 0 || Wo <= 0) return (int)hipErrorInvalidValue;
  const long long Kall = (long long)c->B * Ho * Wo;
  if ((c->C & 7) || (c->O & 7) || (Kall & 31) || c->nchunks > Kall / 32) return (int)hipErrorNotSupported;
  const long long rows_x = (long long)c->B * c->H * c->W + 1;
  if (rows_x * c->C * 2 >= 0xffffffffLL || Kall * c->O * 2 >= 0x7fffffffffffLL) return (int)hipErrorNotSupported;
  KArgs g = {};
  cips_gemm_x3_desc& d = g.d;
  d.M = c->O; d.N = c->C; d.K = (int)(Kall / c->nchunks); d.lda = c->O; d.ldb = c->C; d.batch = c->nchunks;
  d.strideA = 0; d.strideB = 0;                 // chunk starts come from (ktiles, nchunks) in the kernel
  d.ldc = c->C; d.strideC = (long long)c->kh * c->kw * c->O * c->C;
  g.A_hi[0] = c->dy_hi; g.A_lo[0] = c->dy_lo; g.B_hi[0] = c->x_hi; g.B_lo[0] = c->x_lo; g.C[0] = c->part;
  g.cv.C = c->C; g.cv.H = c->H; g.cv.W = c->W; g.cv.kw = c->kw; g.cv.stride = c->stride; g.cv.pad = c->pad;
  g.cv.Ho = Ho; g.cv.Wo = Wo; g.cv.ntap = c->kh * c->kw; g.cv.zero_row = rows_x - 1; g.cv.ktiles = (int)(Kall / 32); g.cv.nchunks = c->nchunks;
  g.tiles_m = (d.M + BM - 1) / BM;
  g.tiles_n = (d.N + BN - 1) / BN;
  g.ngroups = g.cv.ntap;
  const long long total = (long long)g.tiles_m * g.tiles_n * d.batch * g.ngroups;
  if (total > 0x7fffffffLL) return (int)hipErrorInvalidValue;
  g.total = (int)total;
  static int ncu = 0;
  CIPS_PER_DEVICE(ncu, 0);
  if (!ncu) {
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
    if (ncu <= 0) ncu = 256;
    ncu = (ncu / 8) * 8;
    (void)hipFuncSetAttribute((const void*)gemm_bf16x3_km_wide_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm_bf16x3_km_v3_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
  }
  const int grid = g.total < ncu ? g.total : ncu;
  if (g.cv.ktiles / g.cv.nchunks >= 2)          // every chunk has at least two k-tiles (else: the one-k-tile form)
    hipLaunchKernelGGL(gemm_bf16x3_km_v3_kernel<true>, dim3(grid), dim3(512), SMEM_BYTES, (hipStream_t)stream, g);
  else
    hipLaunchKernelGGL(gemm_bf16x3_km_wide_kernel<true>, dim3(grid), dim3(512), SMEM_BYTES, (hipStream_t)stream, g);
  return CIPS_CHECK_LAUNCH();
}
