// gemm_bf16x3.hip — fp32-grade batched GEMM on the bf16 matrix cores of gfx950 by 3-pass
// operand splitting ("bf16x3"), for the CIPS INR head (H4), where 87 % of the generator's
// FLOPs live (17 modulated 512x512 layers x {fwd, dX, dW}).
//
// Every fp32 operand x is carried as two bf16 planes, x = hi + lo (hi = rne_bf16(x),
// lo = rne_bf16(x - hi); |x - hi - lo| <= 2^-17 |x|).  A product sum is evaluated as
//     sum a*b  ~=  sum a_hi*b_hi + a_hi*b_lo + a_lo*b_hi        (dropped a_lo*b_lo ~ 2^-18)
// with three v_mfma_f32_32x32x16_bf16 per fragment pair, accumulating in fp32.  Per-layer relative
// error ~1e-5 against 1e-3 allowed by the parity bar (SURVEY.md §7 "3-pass split-bf16 scheme");
// the matrix pipe runs 16x the fp32-MFMA rate, so three passes = 5.3x the fp32 MFMA roof
// (~0.83 PFLOP/s effective).  The SIREN keeps exact fp32 MFMA (its FiLM gains amplify phase error).
//
// One kernel form, "NT": C[m][n] = sum_k A[m][k] * B[n][k], both operands with the contraction index
// contiguous (what the bf16 MFMA fragments want: 8 consecutive k per lane = one 16-byte LDS read).
// All three INR GEMMs are expressed in it by keeping every activation / gradient in HBM in BOTH
// orientations (row-major planes and transposed planes), written by the producing GEMM's epilogue:
//   forward  Y  = X  . Wbt^T      A = X   [rows][in]     B = Wbt [out][in]
//   dX       dX = G  . Wb^T       A = G   [rows][out]    B = Wb  [in][out]
//   dW       dW = XT . GT^T       A = XT  [in][rows]     B = GT  [out][rows]
//
// Tiling (wave64): 256x128x32 workgroup tile, 512 threads = 8 waves as 4(M) x 2(N), each wave a
// 64x64 sub-tile = 2x2 MFMA tiles of 32x32 (64 accumulator VGPRs), 24 MFMAs per k-tile per wave.
// Staging is LDS-DMA (global_load_lds_dwordx4, no VGPR round trip): four planes per stage (A_hi, A_lo,
// B_hi, B_lo), 64-byte rows without padding, the four 16-byte k-chunks of a row XOR-swizzled by
// (row>>2)&3 — applied to the per-lane SOURCE address of the DMA (its LDS image is lane-linear) and to the
// ds_read_b128 fragment reads, which makes both conflict-free.  A 3-stage ring keeps two k-tiles of DMA in
// flight across raw s_barriers with counted s_waitcnt vmcnt(N); one barrier per k-tile.  Workgroups are
// persistent (one per CU) and walk the tile list XCD-contiguously.  The epilogue stages every auxiliary
// input and every output through a per-wave LDS scratch so that HBM only sees 16-byte accesses.
// A 128x128 / 4-wave / 2-stage form (two workgroups per CU) is selectable with CIPS_X3_TILE=128
// (measured equal within noise).
#include "common.h"
#include "../../include/cips3d_hip.h"
#include <stdlib.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

constexpr int BN = 128, BK = 32;
constexpr int ROWB = 64;                     // LDS row pitch in bytes: 32 bf16, no padding; the four 16-byte k-chunks
                                             // of a row are XOR-swizzled by (row>>2)&3 so that both the DMA image
                                             // (lane-linear) and the ds_read_b128 fragment reads are conflict-free
// Workgroup tile (64*WM) x 128: WM = 4 -> 256x128, 8 waves, 1 workgroup per CU (120 KB LDS);
//                               WM = 2 -> 128x128, 4 waves, 2 workgroups per CU (80 KB LDS each), so one
// workgroup's barriers / prologue / epilogue overlap the other's MFMA phase.
template <int WM> struct Cfg {
  static constexpr int BM = 64 * WM;
  static constexpr int THREADS = WM * 2 * 64;
  static constexpr int NB = 8 / (2 * WM);               // B row groups (16 rows) per wave per plane
  static constexpr int OFF_AHI = 0;
  static constexpr int OFF_ALO = OFF_AHI + BM * ROWB;
  static constexpr int OFF_BHI = OFF_ALO + BM * ROWB;
  static constexpr int OFF_BLO = OFF_BHI + BN * ROWB;
  static constexpr int STAGE = OFF_BLO + BN * ROWB;
  static constexpr int NSTAGE = (WM == 4) ? 3 : 2;      // LDS ring depth: 256-row form 3 stages (1 WG / CU); 128-row form 2 stages, 72 KB -> 2 WGs / CU
  static constexpr int PIECES = 4 + 2 * NB;             // LDS-DMA instructions per wave per k-tile
  static constexpr int SMEM_EPI = WM * 2 * 2 * 64 * 72 * 2;   // per-wave epilogue scratch
  static constexpr int SMEM_BYTES = (NSTAGE * STAGE > SMEM_EPI) ? NSTAGE * STAGE : SMEM_EPI;
};

struct Args {
  cips_gemm_x3_desc d;
  int tiles_m, tiles_n, total;
  int stagger_cycles;   // start-phase quantum (shader cycles), 0 = no staggering
  int ncu;
  int dbg;              // tuning only (env CIPS_X3_GDBG): bit0 skip fragment reads + MFMA, bit1 skip the LDS-DMA loads, bit2 skip the main loop
};

__device__ __forceinline__ u16 f2bf(float v) {
  unsigned u = __float_as_uint(v);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (u16)(u >> 16);
}
__device__ __forceinline__ float bf2f(u16 h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ void split2(float v, u16& hi, u16& lo) {
  hi = f2bf(v);
  lo = f2bf(v - bf2f(hi));
}

template <int WM>
__global__ __launch_bounds__(Cfg<WM>::THREADS, 2) void gemm_bf16x3_kernel(Args g) {
  using CF = Cfg<WM>;
  constexpr int BM = CF::BM, OFF_AHI = CF::OFF_AHI, OFF_ALO = CF::OFF_ALO, OFF_BHI = CF::OFF_BHI, OFF_BLO = CF::OFF_BLO,
                STAGE = CF::STAGE, NB = CF::NB;
  (void)BM;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const cips_gemm_x3_desc& d = g.d;

  // Persistent workgroups: the grid is one workgroup per CU (or fewer); each walks the tile list with a
  // stride of gridDim.x, which keeps it on its XCD (gridDim.x % 8 == 0) and saves the per-tile workgroup
  // launch / LDS (re)allocation latency (~3 us against ~25 us of main loop at K = 512).
  // Optional de-phasing of the persistent workgroups (CIPS_X3_STAGGER=1): four start phases, so that not every
  // CU reaches its store-heavy epilogue at the same moment.  Measured neutral on MI355X: the epilogue traffic adds
  // to the main loop's time whatever the phase relation (268 MB of fp32 C cost ~75 us on top of a 205 us main
  // loop), i.e. the kernel is bound by the memory system, not by MFMA issue.
  if (CIPS_TUNE(g.stagger_cycles) > 0) {
    // 256-row form: four phases across CUs.  128-row form (two workgroups per CU): the second half of the grid
    // (the co-resident partner of workgroup b is b + #CUs) starts half a tile later, so that one workgroup's
    // store drain overlaps its partner's MFMA phase.
    const int phase = (WM == 4) ? ((blockIdx.x >> 3) & 3) : (blockIdx.x >= g.ncu ? 2 : 0);
    for (int i = 0; i < phase * g.stagger_cycles; i += 64 * 100) __builtin_amdgcn_s_sleep(100);
  }
  for (int tseq = blockIdx.x; tseq < g.total; tseq += gridDim.x) {
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
  const int bz = bid / (g.tiles_n * g.tiles_m);
  const int m0 = tm * BM, n0 = tn * BN;
  const int M = d.M, N = d.N, K = d.K;

  const u16* Ahi = (const u16*)d.A_hi + (long long)bz * d.strideA;
  const u16* Alo = (const u16*)d.A_lo + (long long)bz * d.strideA;
  const u16* Bhi = (const u16*)d.B_hi + (long long)bz * d.strideB;
  const u16* Blo = (const u16*)d.B_lo + (long long)bz * d.strideB;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hf = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // HBM -> LDS staging by LDS-DMA (global_load_lds_dwordx4): one wave instruction moves 16 rows x 64 B of one
  // plane; its LDS image is lane-linear (base + lane*16), so lane L = (row L>>2, slot L&3) fetches the global
  // chunk kc = slot ^ ((row>>2)&3) — the swizzle lives in the SOURCE address, the reads apply the same XOR.
  // Rows past M / N are clamped (their products only reach outputs that are never stored); K % 32 == 0.
  // Two LDS stages: the DMA for tile t+1 runs while tile t is multiplied; one barrier per k-tile.
  // The DMA is issued in the SGPR-base form (asm, see gemm_bf16x3_wide.hip): uniform plane pointer of the k-tile + one
  // 32-bit byte offset per lane and row group, computed once per output tile.
  const int drow = lane >> 2, dslot = lane & 3;
  const int uw = __builtin_amdgcn_readfirstlane(wave);
  constexpr int NW = 2 * WM;
  const unsigned sbase_nt = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)smem);
  auto dma_s = [&](const u16* p, unsigned off, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(off), "s"(p), "s"(lds_addr) : "memory");
  };
  auto row_off = [&](int ld, int rows_total, int row_base) -> unsigned {
    const int row = row_base + drow;                                  // row inside the workgroup tile
    const int kcsw = dslot ^ ((row >> 2) & 3);
    const int grow = (row < rows_total) ? row : rows_total - 1;
    return (unsigned)(grow * ld + kcsw * 8) * 2u;
  };
  unsigned offA_nt[2], offB_nt[NB];
#pragma unroll
  for (int i = 0; i < 2; ++i) offA_nt[i] = row_off(d.lda, M - m0, (uw + NW * i) * 16);
#pragma unroll
  for (int i = 0; i < NB; ++i) offB_nt[i] = row_off(d.ldb, N - n0, (uw + NW * i) * 16);
  auto issue_tile = [&](int stage, int k0) {
    const unsigned s = sbase_nt + stage * STAGE;
    const u16 *ah = Ahi + (long long)m0 * d.lda + k0, *al = Alo + (long long)m0 * d.lda + k0;
    const u16 *bh = Bhi + (long long)n0 * d.ldb + k0, *bl = Blo + (long long)n0 * d.ldb + k0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {            // A: BM/16 row groups per plane, 2 per wave
      const int gidx = uw + NW * i;
      dma_s(ah, offA_nt[i], s + OFF_AHI + gidx * 16 * ROWB);
      dma_s(al, offA_nt[i], s + OFF_ALO + gidx * 16 * ROWB);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {           // B: 8 row groups per plane
      const int gidx = uw + NW * i;
      dma_s(bh, offB_nt[i], s + OFF_BHI + gidx * 16 * ROWB);
      dma_s(bl, offB_nt[i], s + OFF_BLO + gidx * 16 * ROWB);
    }
  };
  auto compute = [&](int stage) {
    const unsigned char* s = smem + stage * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int R = wm * 64 + i * 32 + l31;
        const int off = R * ROWB + (((ks * 2 + hf) ^ ((R >> 2) & 3)) << 4);
        ah[i] = *reinterpret_cast<const bf16x8*>(s + OFF_AHI + off);
        al[i] = *reinterpret_cast<const bf16x8*>(s + OFF_ALO + off);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int R = wn * 64 + j * 32 + l31;
        const int off = R * ROWB + (((ks * 2 + hf) ^ ((R >> 2) & 3)) << 4);
        bh[j] = *reinterpret_cast<const bf16x8*>(s + OFF_BHI + off);
        bl[j] = *reinterpret_cast<const bf16x8*>(s + OFF_BLO + off);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
  };

  // Ring of NSTAGE LDS stages, NSTAGE-1 tiles of LDS-DMA in flight across the (raw) barriers: the wait in
  // front of tile kt is a COUNTED vmcnt that leaves the younger tiles' pieces outstanding
  // (cdna_hip_programming.md T3/T4: never drain to 0 in the main loop).
  constexpr int NSTAGE = CF::NSTAGE, PIECES = CF::PIECES, DIST = NSTAGE - 1;
  const int nk = CIPS_TUNE(g.dbg & 4) ? 0 : K / BK;               // bit2: epilogue only
#pragma unroll
  for (int t = 0; t < DIST; ++t)
    if (t < nk && !CIPS_TUNE(g.dbg & 2)) issue_tile(t, t * BK);
  for (int kt = 0; kt < nk; ++kt) {
    const int younger = min(DIST - 1, nk - 1 - kt);      // tiles issued after kt that may stay in flight
    if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PIECES) : "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                         // every wave's pieces of tile kt have landed;
                                                          // every wave is done reading stage (kt-1) % NSTAGE
    if (kt + DIST < nk && !CIPS_TUNE(g.dbg & 2)) issue_tile((kt + DIST) % NSTAGE, (kt + DIST) * BK);
    if (!CIPS_TUNE(g.dbg & 1)) compute(kt % NSTAGE);
  }
  __builtin_amdgcn_s_barrier();

  // ---------------- epilogue ----------------
  // Every auxiliary tensor enters and every result leaves the CU through a per-wave LDS scratch tile
  // (64x64, pitch 72) with 16-byte global accesses; the per-lane MFMA-layout values only ever touch LDS.
  const long long cb = (long long)bz * d.strideC;
  const long long pb = (long long)bz * d.strideP;
  const long long tb = (long long)bz * d.strideT;
  u16* Phi = (u16*)d.P_hi; u16* Plo = (u16*)d.P_lo;
  u16* Thi = (u16*)d.T_hi; u16* Tlo = (u16*)d.T_lo;
  constexpr int PITCH = 72;
  u16* sc_hi = reinterpret_cast<u16*>(smem) + wave * (2 * 64 * PITCH);
  u16* sc_lo = sc_hi + 64 * PITCH;
  float* sc_f = reinterpret_cast<float*>(sc_hi);               // same bytes viewed as one fp32 [64][72] image
  const int wrow0 = m0 + wm * 64, wcol0 = n0 + wn * 64;
  const bool vec_c = ((d.ldc & 3) == 0) && ((d.strideC & 3) == 0);
  const bool vec_p = ((d.ldp & 7) == 0) && ((d.strideP & 7) == 0);
  const bool vec_t = ((d.ldt & 7) == 0) && ((d.strideT & 7) == 0);

  float v[2][2][16];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) v[i][j][r] = acc[i][j][r];

  // fp32 [M][ldc] tile <-> scratch (1024 float4 chunks per wave tile)
  auto tile_in_f32 = [&](const float* base) {
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const int id = lane + 64 * c, rr = id >> 4, c4 = (id & 15) * 4;
      const int row = wrow0 + rr, col = wcol0 + c4;
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < M && col < N) {
        const float* p = base + cb + (long long)row * d.ldc + col;
        if (vec_c && col + 3 < N) t = *reinterpret_cast<const float4*>(p);
        else { t.x = p[0]; if (col + 1 < N) t.y = p[1]; if (col + 2 < N) t.z = p[2]; if (col + 3 < N) t.w = p[3]; }
      }
      *reinterpret_cast<float4*>(sc_f + rr * PITCH + c4) = t;
    }
  };
  auto tile_out_f32 = [&](float* base) {
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const int id = lane + 64 * c, rr = id >> 4, c4 = (id & 15) * 4;
      const int row = wrow0 + rr, col = wcol0 + c4;
      if (row < M && col < N) {
        const float4 t = *reinterpret_cast<const float4*>(sc_f + rr * PITCH + c4);
        float* p = base + cb + (long long)row * d.ldc + col;
        if (vec_c && col + 3 < N) *reinterpret_cast<float4*>(p) = t;
        else { p[0] = t.x; if (col + 1 < N) p[1] = t.y; if (col + 2 < N) p[2] = t.z; if (col + 3 < N) p[3] = t.w; }
      }
    }
  };
  // bf16 [M][ldp] tile <-> one scratch image (512 chunks of 8)
  auto tile_in_bf16 = [&](const u16* base, u16* img) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int id = lane + 64 * c, rr = id >> 3, c8 = (id & 7) * 8;
      const int row = wrow0 + rr, col = wcol0 + c8;
      uint4 t = make_uint4(0, 0, 0, 0);
      if (row < M && col < N) {
        const u16* p = base + pb + (long long)row * d.ldp + col;
        if (vec_p && col + 7 < N) t = *reinterpret_cast<const uint4*>(p);
        else {
          u16 e[8] = {0, 0, 0, 0, 0, 0, 0, 0};
          for (int q = 0; q < 8 && col + q < N; ++q) e[q] = p[q];
          t = make_uint4(e[0] | ((unsigned)e[1] << 16), e[2] | ((unsigned)e[3] << 16), e[4] | ((unsigned)e[5] << 16),
                         e[6] | ((unsigned)e[7] << 16));
        }
      }
      *reinterpret_cast<uint4*>(img + rr * PITCH + c8) = t;
    }
  };
  auto tile_out_bf16 = [&](u16* base, const u16* img) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int id = lane + 64 * c, rr = id >> 3, c8 = (id & 7) * 8;
      const int row = wrow0 + rr, col = wcol0 + c8;
      if (row < M && col < N) {
        u16* p = base + pb + (long long)row * d.ldp + col;
        if (vec_p && col + 7 < N) *reinterpret_cast<uint4*>(p) = *reinterpret_cast<const uint4*>(img + rr * PITCH + c8);
        else for (int q = 0; q < 8 && col + q < N; ++q) p[q] = img[rr * PITCH + c8 + q];
      }
    }
  };
// The scratch regions are private to a wave and a wave's DS instructions execute in order, so a
// write -> read hand-over inside one wave needs no workgroup barrier: only the compiler must keep the order
// and the LDS queue must be drained before registers are reused.
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                         __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#define CIPS_FOR_ELEMS(BODY)                                              \
  _Pragma("unroll") for (int i = 0; i < 2; ++i)                            \
  _Pragma("unroll") for (int j = 0; j < 2; ++j)                            \
  _Pragma("unroll") for (int r = 0; r < 16; ++r) {                         \
    const int so = (i * 32 + mfma_row(r, hf)) * PITCH + j * 32 + l31;      \
    BODY                                                                   \
  }

  __syncthreads();   // main-loop LDS reads are done everywhere; scratch regions are per wave from here on
  if (d.add) {
    tile_in_f32(d.add);
    WAVE_SYNC();
    CIPS_FOR_ELEMS(v[i][j][r] += sc_f[so];)
    WAVE_SYNC();
  }
  if (d.rgb_g) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = wcol0 + j * 32 + l31;
      float rw0 = 0.f, rw1 = 0.f, rw2 = 0.f;
      if (col < N) { rw0 = d.rgb_w[col]; rw1 = d.rgb_w[N + col]; rw2 = d.rgb_w[2 * N + col]; }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wrow0 + i * 32 + mfma_row(r, hf);
          if (row < M) {
            const float* gp = d.rgb_g + ((long long)bz * M + row) * 3;
            v[i][j][r] = fmaf(gp[0], rw0, fmaf(gp[1], rw1, fmaf(gp[2], rw2, v[i][j][r])));
          }
        }
    }
  }
  if (d.C_unmasked) {
    CIPS_FOR_ELEMS(sc_f[so] = v[i][j][r];)
    WAVE_SYNC();
    tile_out_f32(d.C_unmasked);
    WAVE_SYNC();
  }
  if (d.mask && (d.gate_bits & 1)) {
    // bit plane: the 32 columns of an MFMA tile row are one dword (N, ldp, strideP multiples of 32), bit = lane & 31
    const unsigned char* mbase = (const unsigned char*)d.mask + (pb >> 3);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wrow0 + i * 32 + mfma_row(r, hf), colb = wcol0 + j * 32;
          unsigned w = 0;
          if (row < M && colb < N) w = *reinterpret_cast<const unsigned*>(mbase + (((long long)row * d.ldp + colb) >> 3));
          v[i][j][r] *= ((w >> l31) & 1u) ? 1.f : d.slope;
        }
  } else if (d.mask) {
    tile_in_bf16((const u16*)d.mask, sc_hi);
    WAVE_SYNC();
    CIPS_FOR_ELEMS(
      const u16 mb = sc_hi[so];
      const bool pos = ((mb & 0x8000u) == 0) && ((mb & 0x7fffu) != 0);
      v[i][j][r] *= pos ? 1.f : d.slope;)
    WAVE_SYNC();
  }
  if (d.act) { CIPS_FOR_ELEMS(v[i][j][r] = lrelu(v[i][j][r], d.slope); (void)so;) }
  if (d.mask_out && (d.gate_bits & 2)) {
    unsigned char* obase = (unsigned char*)d.mask_out + (pb >> 3);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const unsigned long long bal = __ballot(v[i][j][r] > 0.f);      // lanes 0..31: row of hf 0, 32..63: row of hf 1
          const int row = wrow0 + i * 32 + mfma_row(r, hf), colb = wcol0 + j * 32;
          if (l31 == 0 && row < M && colb < N)
            *reinterpret_cast<unsigned*>(obase + (((long long)row * d.ldp + colb) >> 3)) = (unsigned)(hf ? (bal >> 32) : bal);
        }
  } else if (d.mask_out) {
    CIPS_FOR_ELEMS(sc_hi[so] = f2bf(v[i][j][r]);)
    WAVE_SYNC();
    tile_out_bf16((u16*)d.mask_out, sc_hi);
    WAVE_SYNC();
  }
  if (d.res_hi) {
    tile_in_bf16((const u16*)d.res_hi, sc_hi);
    tile_in_bf16((const u16*)d.res_lo, sc_lo);
    WAVE_SYNC();
    CIPS_FOR_ELEMS(v[i][j][r] += bf2f(sc_hi[so]) + bf2f(sc_lo[so]);)
    WAVE_SYNC();
  }
  if (d.C) {
    CIPS_FOR_ELEMS(sc_f[so] = v[i][j][r];)
    WAVE_SYNC();
    tile_out_f32(d.C);
    WAVE_SYNC();
  }
  if (Phi || Thi) {
    u16 vh[2][2][16], vl[2][2][16];
    CIPS_FOR_ELEMS(split2(v[i][j][r], vh[i][j][r], vl[i][j][r]); (void)so;)
    if (Phi) {
      CIPS_FOR_ELEMS(sc_hi[so] = vh[i][j][r]; sc_lo[so] = vl[i][j][r];)
      WAVE_SYNC();
      tile_out_bf16(Phi, sc_hi);
      tile_out_bf16(Plo, sc_lo);
      WAVE_SYNC();
    }
    if (Thi) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            // 4 consecutive rows (i*32 + 8*rg + 4*hf + 0..3) of column j*32+l31 -> one 8-byte LDS write
            const int o = (j * 32 + l31) * PITCH + i * 32 + 8 * rg + 4 * hf;
            *reinterpret_cast<uint2*>(sc_hi + o) = make_uint2(vh[i][j][4 * rg] | ((unsigned)vh[i][j][4 * rg + 1] << 16),
                                                              vh[i][j][4 * rg + 2] | ((unsigned)vh[i][j][4 * rg + 3] << 16));
            *reinterpret_cast<uint2*>(sc_lo + o) = make_uint2(vl[i][j][4 * rg] | ((unsigned)vl[i][j][4 * rg + 1] << 16),
                                                              vl[i][j][4 * rg + 2] | ((unsigned)vl[i][j][4 * rg + 3] << 16));
          }
      WAVE_SYNC();
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int id = lane + 64 * c;
        const int cc = id >> 3, r8 = (id & 7) * 8;      // transposed image: row = output column, 8 consecutive m
        const int col = wcol0 + cc, row = wrow0 + r8;
        if (col < N && row < M) {
          const long long o = tb + (long long)col * d.ldt + row;
          if (vec_t && row + 7 < M) {
            *reinterpret_cast<uint4*>(Thi + o) = *reinterpret_cast<const uint4*>(sc_hi + cc * PITCH + r8);
            *reinterpret_cast<uint4*>(Tlo + o) = *reinterpret_cast<const uint4*>(sc_lo + cc * PITCH + r8);
          } else {
            for (int e = 0; e < 8 && row + e < M; ++e) {
              Thi[o + e] = sc_hi[cc * PITCH + r8 + e]; Tlo[o + e] = sc_lo[cc * PITCH + r8 + e];
            }
          }
        }
      }
    }
  }
  __syncthreads();   // scratch is free again before the next tile's LDS-DMA lands in it
  }  // persistent tile loop
#undef CIPS_FOR_ELEMS
#undef WAVE_SYNC
}

// ------------------------------------------------------------------------------------------------
// K-major form ("TN"):  C[m][n] = sum_k A[k][m] * B[k][n], both operands stored with the contraction
// index as the ROW (A planes [K][lda], B planes [K][ldb]) — i.e. the row-major activation / gradient planes
// themselves, so the weight-gradient GEMMs (dW = X^T G) need no transposed copies in HBM.
// The MFMA fragments want 8 consecutive k per lane; they are produced by the LDS transpose read
// ds_read_b64_tr_b16 (semantics probed on hardware, scripts/probe/tr_probe.hip: within a 16-lane group lane t
// receives, for j = 0..3, element (t & 3) of the 8-byte chunk addressed by lane 4j + (t >> 2)): lane s of a
// group points at row kb + (s>>2), columns mb + 4(s&3).. of the k-major tile and gets column mb + s, rows
// kb..kb+3.  LDS image per stage: A [32 k][256 m], B [32 k][128 n] bf16 per plane, rows unpadded; 32-byte
// column pairs XOR-swizzled by 2*(k&3) (in the DMA source address and in the reads) so that the 32 lanes of a
// service group hit 8 distinct 32-byte segments.  Pipeline, ring and persistence as in the NT kernel.
// ------------------------------------------------------------------------------------------------
typedef short short4v __attribute__((ext_vector_type(4)));
typedef short short8v __attribute__((ext_vector_type(8)));

template <int WM>   // WM = 4: 256x128 tile, 8 waves, 3-stage ring (1 WG/CU); WM = 2: 128x128 tile, 4 waves, 2 stages (2 WGs/CU)
struct KmCfg {
  static constexpr int BM = 64 * WM, NW = 2 * WM, THREADS = 64 * NW;
  static constexpr int NSTAGE = (WM == 4) ? 3 : 2;
  static constexpr int AROW = 2 * BM, BROW = 256;                       // bytes per k-row of the A / B image
  static constexpr int A_ROWS_PER_PIECE = 1024 / AROW;                  // 2 (WM=4) or 4 (WM=2)
  static constexpr int A_PIECES = 32 / A_ROWS_PER_PIECE;                // per plane: 16 or 8
  static constexpr int A_PER_WAVE = A_PIECES / NW;                      // 2
  static constexpr int B_PER_WAVE = 8 / NW;                             // 1 or 2
  static constexpr int PIECES = 2 * A_PER_WAVE + 2 * B_PER_WAVE;        // 6 or 8
  static constexpr int OFF_AHI = 0, OFF_ALO = 32 * AROW, OFF_BHI = 2 * 32 * AROW, OFF_BLO = OFF_BHI + 32 * BROW;
  static constexpr int STAGE = OFF_BLO + 32 * BROW;                     // 49152 or 32768
  static constexpr int SMEM_EPI = NW * 64 * 72 * 4;
  static constexpr int SMEM_BYTES = (NSTAGE * STAGE > SMEM_EPI) ? NSTAGE * STAGE : SMEM_EPI;
};

template <int WM>
__global__ __launch_bounds__(KmCfg<WM>::THREADS, 2) void gemm_bf16x3_km_kernel(Args g) {
  using KC = KmCfg<WM>;
  constexpr int BM = KC::BM, NSTAGE = KC::NSTAGE, PIECES = KC::PIECES, DIST = NSTAGE - 1, NW = KC::NW;
  constexpr int AROW = KC::AROW, BROW = KC::BROW;
  constexpr int OFF_AHI = KC::OFF_AHI, OFF_ALO = KC::OFF_ALO, OFF_BHI = KC::OFF_BHI, OFF_BLO = KC::OFF_BLO;
  constexpr int STAGE = KC::STAGE;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const cips_gemm_x3_desc& d = g.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hf = lane >> 5;
  const int s16 = lane & 15, mhalf = (lane >> 4) & 1;
  const int uw = __builtin_amdgcn_readfirstlane(wave);

  for (int tseq = blockIdx.x; tseq < g.total; tseq += gridDim.x) {
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
  const int bz = bid / (g.tiles_n * g.tiles_m);
  const int m0 = tm * BM, n0 = tn * BN;
  const int M = d.M, N = d.N, K = d.K;
  const u16* Ahi = (const u16*)d.A_hi + (long long)bz * d.strideA;
  const u16* Alo = (const u16*)d.A_lo + (long long)bz * d.strideA;
  const u16* Bhi = (const u16*)d.B_hi + (long long)bz * d.strideB;
  const u16* Blo = (const u16*)d.B_lo + (long long)bz * d.strideB;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // LDS-DMA in the SGPR-base form (asm, see gemm_bf16x3_wide.hip): uniform row pointer of the k-tile + one 32-bit
  // byte offset per lane and piece, computed once per output tile
  const unsigned sbase_km = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)smem);
  auto dma_s = [&](const u16* p, unsigned off, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(off), "s"(p), "s"(lds_addr) : "memory");
  };
  unsigned offA_km[KC::A_PER_WAVE], offB_km[KC::B_PER_WAVE];
#pragma unroll
  for (int i = 0; i < KC::A_PER_WAVE; ++i) {             // A: A_PIECES 1-KiB pieces per plane
    const int idx = uw + NW * i;
    constexpr int LPR = 64 / KC::A_ROWS_PER_PIECE;       // lanes per k-row of a piece
    const int k = KC::A_ROWS_PER_PIECE * idx + lane / LPR, c16 = lane % LPR;
    const int pr = (c16 >> 1) ^ (2 * (k & 3));
    int m = m0 + (pr * 2 + (c16 & 1)) * 8;
    m = (m < M) ? m : 0;                                 // clamped columns only feed outputs that are never stored
    offA_km[i] = (unsigned)(k * d.lda + m) * 2u;
  }
#pragma unroll
  for (int i = 0; i < KC::B_PER_WAVE; ++i) {             // B: 8 pieces per plane (4 k-rows each)
    const int idx = uw + NW * i;
    const int k = 4 * idx + (lane >> 4), c16 = lane & 15;
    const int pr = (c16 >> 1) ^ (2 * (k & 3));
    int n = n0 + (pr * 2 + (c16 & 1)) * 8;
    n = (n < N) ? n : 0;
    offB_km[i] = (unsigned)(k * d.ldb + n) * 2u;
  }
  auto issue_tile = [&](int stage, int k0) {
    const unsigned s = sbase_km + stage * STAGE;
    const u16 *ah = Ahi + (long long)k0 * d.lda, *al = Alo + (long long)k0 * d.lda;
    const u16 *bh = Bhi + (long long)k0 * d.ldb, *bl = Blo + (long long)k0 * d.ldb;
#pragma unroll
    for (int i = 0; i < KC::A_PER_WAVE; ++i) {
      const int idx = uw + NW * i;
      dma_s(ah, offA_km[i], s + OFF_AHI + idx * 1024);
      dma_s(al, offA_km[i], s + OFF_ALO + idx * 1024);
    }
#pragma unroll
    for (int i = 0; i < KC::B_PER_WAVE; ++i) {
      const int idx = uw + NW * i;
      dma_s(bh, offB_km[i], s + OFF_BHI + idx * 1024);
      dma_s(bl, offB_km[i], s + OFF_BLO + idx * 1024);
    }
  };
  // transpose-read one 32(m) x 16(k) fragment: two ds_read_b64_tr_b16 (rows kb.. and kb+4..)
  auto frag = [&](const unsigned char* plane, int rowbytes, int col0, int ks) -> bf16x8 {
    const int kb = 16 * ks + 8 * hf + (s16 >> 2);                     // this lane's source row for the first read
    const int col = col0 + 16 * mhalf + 4 * (s16 & 3);
    const int pr = (col >> 4) ^ (2 * (kb & 3));                       // (kb+4)&3 == kb&3
    const unsigned char* p0 = plane + kb * rowbytes + pr * 32 + (col & 15) * 2;
    short4v a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)p0);
    short4v b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(p0 + 4 * rowbytes));
    short8v v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  auto compute = [&](int stage) {
    const unsigned char* s = smem + stage * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ah[i] = frag(s + OFF_AHI, AROW, wm * 64 + i * 32, ks);
        al[i] = frag(s + OFF_ALO, AROW, wm * 64 + i * 32, ks);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        bh[j] = frag(s + OFF_BHI, BROW, wn * 64 + j * 32, ks);
        bl[j] = frag(s + OFF_BLO, BROW, wn * 64 + j * 32, ks);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
  };

  const int nk = K / BK;
#pragma unroll
  for (int t = 0; t < DIST; ++t)
    if (t < nk) issue_tile(t, t * BK);
  for (int kt = 0; kt < nk; ++kt) {
    const int younger = min(DIST - 1, nk - 1 - kt);
    if (younger >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + DIST < nk) issue_tile((kt + DIST) % NSTAGE, (kt + DIST) * BK);
    compute(kt % NSTAGE);
  }
  __builtin_amdgcn_s_barrier();

  // ---- epilogue: fp32 C through the per-wave LDS scratch, 16-byte stores ----
  constexpr int PITCH = 72;
  float* sc_f = reinterpret_cast<float*>(smem) + wave * (64 * PITCH);
  const int wrow0 = m0 + wm * 64, wcol0 = n0 + wn * 64;
  const long long cb = (long long)bz * d.strideC;
  const bool vec_c = ((d.ldc & 3) == 0) && ((d.strideC & 3) == 0);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) sc_f[(i * 32 + mfma_row(r, hf)) * PITCH + j * 32 + l31] = acc[i][j][r];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const int id = lane + 64 * c, rr = id >> 4, c4 = (id & 15) * 4;
    const int row = wrow0 + rr, col = wcol0 + c4;
    if (row < M && col < N) {
      const float4 t = *reinterpret_cast<const float4*>(sc_f + rr * PITCH + c4);
      float* p = d.C + cb + (long long)row * d.ldc + col;
      if (vec_c && col + 3 < N) *reinterpret_cast<float4*>(p) = t;
      else { p[0] = t.x; if (col + 1 < N) p[1] = t.y; if (col + 2 < N) p[2] = t.z; if (col + 3 < N) p[3] = t.w; }
    }
  }
  __syncthreads();
  }  // persistent tile loop
}

// fp32 (rows, cols) row-major -> split planes row-major [rows][ldp] and/or transposed [cols][ldt]
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, u16* __restrict__ phi,
                                                           u16* __restrict__ plo, u16* __restrict__ thi,
                                                           u16* __restrict__ tlo, int rows, int cols, int ldx,
                                                           int ldp, int ldt, long long sx, long long sp, long long st) {
  __shared__ u16 th[32][33], tl[32][33];
  const int bz = blockIdx.z;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  for (int rr = ty; rr < 32; rr += 8) {
    const int r = r0 + rr, c = c0 + tx;
    u16 h = 0, l = 0;
    if (r < rows && c < cols) {
      split2(x[bz * sx + (long long)r * ldx + c], h, l);
      if (phi) { phi[bz * sp + (long long)r * ldp + c] = h; plo[bz * sp + (long long)r * ldp + c] = l; }
    }
    th[rr][tx] = h; tl[rr][tx] = l;
  }
  __syncthreads();
  if (thi)
    for (int cc = ty; cc < 32; cc += 8) {
      const int c = c0 + cc, r = r0 + tx;
      if (r < rows && c < cols) {
        thi[bz * st + (long long)c * ldt + r] = th[tx][cc];
        tlo[bz * st + (long long)c * ldt + r] = tl[tx][cc];
      }
    }
}


// NCHW fp32 (B, C, n) -> NHWC split planes (B*n + 1, C) with a zero last row: the operand form of the implicit-GEMM
// convolutions.  64 x 64 tiles: 256-byte read segments along n (float4 per thread), 128-byte write segments along C
// (8 bf16 per thread and plane) — the generic 32 x 32 kernel above writes 64-byte segments and reached 1.9 TB/s on
// the 134 MB activation of the 64 x 64 stage.  Block (0, 0, 0) also writes the zero row (was two torch fills per call).
__global__ __launch_bounds__(256) void split_nhwc_kernel(const float* __restrict__ x, u16* __restrict__ thi, u16* __restrict__ tlo,
                                                         int C, int n, int B) {
  __shared__ __attribute__((aligned(16))) u16 sh[64][72], sl[64][72];      // [n local][c local], 144-byte rows
  const int b = blockIdx.z, c0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int t = threadIdx.x;
  const bool vec = (n & 3) == 0;
  const float* xb = x + (long long)b * C * n;
  {
    const int col4 = t & 15, r_ = t >> 4;                // 16 float4 per 64-float row, 16 rows per pass
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int cl = r_ + 16 * k, c = c0 + cl, nn = n0 + 4 * col4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < C && nn < n) {
        const float* q = xb + (long long)c * n + nn;
        if (vec) v = *reinterpret_cast<const float4*>(q);                   // n % 4 == 0: rows are 16-byte aligned
        else {                                                              // odd planes (65 x 65 blur outputs ...)
          v.x = q[0];
          if (nn + 1 < n) v.y = q[1];
          if (nn + 2 < n) v.z = q[2];
          if (nn + 3 < n) v.w = q[3];
        }
      }
      u16 h, l;
      split2(v.x, h, l); sh[4 * col4 + 0][cl] = h; sl[4 * col4 + 0][cl] = l;
      split2(v.y, h, l); sh[4 * col4 + 1][cl] = h; sl[4 * col4 + 1][cl] = l;
      split2(v.z, h, l); sh[4 * col4 + 2][cl] = h; sl[4 * col4 + 2][cl] = l;
      split2(v.w, h, l); sh[4 * col4 + 3][cl] = h; sl[4 * col4 + 3][cl] = l;
    }
  }
  __syncthreads();
  {
    const int c8 = t & 7, r_ = t >> 3;                   // 8 x 16 B per 64-channel row, 32 rows per pass
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int nl = r_ + 32 * k, nn = n0 + nl, c = c0 + 8 * c8;
      if (nn < n && c < C) {                             // C % 8 == 0
        const long long o = ((long long)b * n + nn) * C + c;
        *reinterpret_cast<uint4*>(thi + o) = *reinterpret_cast<const uint4*>(&sh[nl][8 * c8]);
        *reinterpret_cast<uint4*>(tlo + o) = *reinterpret_cast<const uint4*>(&sl[nl][8 * c8]);
      }
    }
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {
    const long long z = (long long)B * n * C;
    for (int i = t; i < C; i += 256) { thi[z + i] = 0; tlo[z + i] = 0; }
  }
}

// FusedLeakyReLU backward written DIRECTLY as the NHWC split planes of the gated gradient (round 6): for the convolutions that
// read their incoming gradient only through those planes (implicit-GEMM data and weight gradients) the fp32 NCHW tensor
//   gin = (ref > 0 ? g : g * alpha) * scale            (exp/comm/op/fused_act.py:26-44, fused_bias_act_kernel.cu:36-47)
// never exists: one pass reads g and ref (8 B per element) and writes the planes (4 B) — cips_lrelu_bwd_bias (12 B) followed by
// cips_split_planes_nhwc (8 B) moved 20.  Same tile as split_nhwc_kernel (64 channels x 64 pixels, float4 reads along n,
// 16-byte writes along C); per (image, channel, pixel tile) one partial bias sum (the 16 lanes of a channel row reduce by
// shuffles), added over images and tiles by cips_lrelu_bwd_bias_finish.  Plane values are the two kernels' bit for bit.
__global__ __launch_bounds__(256) void lrelu_bwd_nhwc_kernel(const float* __restrict__ g, const float* __restrict__ ref,
                                                             u16* __restrict__ thi, u16* __restrict__ tlo, float* __restrict__ part,
                                                             int C, int n, int B, int ntiles, float alpha, float scale) {
  __shared__ __attribute__((aligned(16))) u16 sh[64][72], sl[64][72];
  const int b = blockIdx.z, c0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int t = threadIdx.x;
  const bool vec = (n & 3) == 0;
  const long long ib = (long long)b * C * n;
  {
    const int col4 = t & 15, r_ = t >> 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int cl = r_ + 16 * k, c = c0 + cl, nn = n0 + 4 * col4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f), r = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < C && nn < n) {
        const long long o = ib + (long long)c * n + nn;
        if (vec) { v = *reinterpret_cast<const float4*>(g + o); r = *reinterpret_cast<const float4*>(ref + o); }
        else {
          v.x = g[o]; r.x = ref[o];
          if (nn + 1 < n) { v.y = g[o + 1]; r.y = ref[o + 1]; }
          if (nn + 2 < n) { v.z = g[o + 2]; r.z = ref[o + 2]; }
          if (nn + 3 < n) { v.w = g[o + 3]; r.w = ref[o + 3]; }
        }
      }
      float4 q;
      q.x = (r.x > 0.f ? v.x : v.x * alpha) * scale; q.y = (r.y > 0.f ? v.y : v.y * alpha) * scale;
      q.z = (r.z > 0.f ? v.z : v.z * alpha) * scale; q.w = (r.w > 0.f ? v.w : v.w * alpha) * scale;
      float acc = (q.x + q.y) + (q.z + q.w);
#pragma unroll
      for (int off = 8; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);        // the 16 lanes of this channel row
      if (col4 == 0 && c < C) part[((long long)b * C + c) * ntiles + blockIdx.x] = acc;
      u16 h, l;
      split2(q.x, h, l); sh[4 * col4 + 0][cl] = h; sl[4 * col4 + 0][cl] = l;
      split2(q.y, h, l); sh[4 * col4 + 1][cl] = h; sl[4 * col4 + 1][cl] = l;
      split2(q.z, h, l); sh[4 * col4 + 2][cl] = h; sl[4 * col4 + 2][cl] = l;
      split2(q.w, h, l); sh[4 * col4 + 3][cl] = h; sl[4 * col4 + 3][cl] = l;
    }
  }
  __syncthreads();
  {
    const int c8 = t & 7, r_ = t >> 3;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int nl = r_ + 32 * k, nn = n0 + nl, c = c0 + 8 * c8;
      if (nn < n && c < C) {
        const long long o = ((long long)b * n + nn) * C + c;
        *reinterpret_cast<uint4*>(thi + o) = *reinterpret_cast<const uint4*>(&sh[nl][8 * c8]);
        *reinterpret_cast<uint4*>(tlo + o) = *reinterpret_cast<const uint4*>(&sl[nl][8 * c8]);
      }
    }
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {
    const long long z = (long long)B * n * C;
    for (int i = t; i < C; i += 256) { thi[z + i] = 0; tlo[z + i] = 0; }
  }
}

// dw[o][c][tap] = scale * sum_chunk part[chunk][tap][o][c]: the tail of the implicit-GEMM weight gradient (was a torch
// reduction, a permuting copy and a scalar multiply per convolution)
__global__ __launch_bounds__(256) void conv_wgrad_finish_kernel(const float* __restrict__ part, float* __restrict__ dw, int nch,
                                                                int taps, long long oc, float scale) {
  const long long total = (long long)taps * oc;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += gridDim.x * 256LL) {
    const int tap = (int)(i / oc);
    const long long j = i - (long long)tap * oc;
    float v = part[i];
    for (int ch = 1; ch < nch; ++ch) v += part[ch * total + i];
    dw[j * taps + tap] = v * scale;
  }
}

}  // namespace

extern "C" CIPS_INTERNAL int cips_gemm_bf16x3_wide(const cips_gemm_x3_desc* d, cips_stream_t stream);   // gemm_bf16x3_wide.hip
extern "C" CIPS_INTERNAL int cips_gemm_bf16x3_v3(const cips_gemm_x3_desc* d, cips_stream_t stream);     // gemm_bf16x3_v3.hip
extern "C" CIPS_INTERNAL int cips_gemm_bf16x3_v3_accepts(const cips_gemm_x3_desc* d);
// Kernel choice (descriptor field `kernel`, cips3d_hip.h): 0 = automatic — 256x256 tiles (v3 schedule for interior shapes,
// else the wide kernel) for problems that fill the chip with them, the 256x128 kernel below otherwise; 1 = the 256x128 kernel
// only; 2 = 256x256 tiles whenever a kernel takes the shape; 3 = like 2 but never the v3 schedule.  1-3 exist for the parity
// tests and microbenchmarks of each kernel; the choice is an argument of the call, the library keeps no mode.
static inline bool x3_big(const cips_gemm_x3_desc* d) {
  return d->N >= 256 && d->M >= 256 && (long long)((d->M + 255) / 256) * ((d->N + 255) / 256) * d->batch >= 256;
}
static inline bool x3_wants_256(const cips_gemm_x3_desc* d) { return d->kernel >= 2 || (d->kernel == 0 && x3_big(d)); }

extern "C" int cips_gemm_bf16x3(const cips_gemm_x3_desc* d, cips_stream_t stream) {
  if (!d || d->M <= 0 || d->N <= 0 || d->K <= 0 || d->batch <= 0) return (int)hipErrorInvalidValue;
  if ((d->K & 31) || (d->lda & 7) || (d->ldb & 7) || (d->strideA & 7) || (d->strideB & 7))
    return (int)hipErrorInvalidValue;
  if (d->kernel < 0 || d->kernel > 3) return (int)hipErrorInvalidValue;
  if (d->gate_bits && ((d->N & 31) || (d->ldp & 31) || (d->strideP & 31))) return (int)hipErrorInvalidValue;
  // large square-ish problems: 256x256 tiles (less operand traffic per flop, prefetched epilogue inputs)
  if (x3_wants_256(d) && d->kernel != 3) {
    const int rc3 = cips_gemm_bf16x3_v3(d, stream);
    if (rc3 != (int)hipErrorNotSupported) return rc3;
  }
  if (d->torgb_w) return (int)hipErrorNotSupported;       // only the v3 kernel folds ToRGB in: never drop it silently
  if (d->addp_hi) return (int)hipErrorNotSupported;       // ... and only it takes the addend as gated planes
  if (x3_wants_256(d)) {
    const int rc = cips_gemm_bf16x3_wide(d, stream);
    if (rc != (int)hipErrorNotSupported) return rc;
  }
  // 256x128 tile / 8 waves / 3-stage ring
  static bool attr = false;
  CIPS_PER_DEVICE(attr, false);
  if (!attr) {
    hipFuncSetAttribute((const void*)gemm_bf16x3_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg<4>::SMEM_BYTES);
    attr = true;
  }
  Args g;
  g.d = *d;
  g.tiles_m = (d->M + 255) / 256;
  g.tiles_n = (d->N + BN - 1) / BN;
  long long total = (long long)g.tiles_m * g.tiles_n * d->batch;
  if (total > 0x7fffffffLL) return (int)hipErrorInvalidValue;
  g.total = (int)total;
  static int ncu = 0;
  CIPS_PER_DEVICE(ncu, 0);
  if (!ncu) {   // persistent grid: one workgroup per CU
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
    if (ncu <= 0) ncu = 256;
    ncu = (ncu / 8) * 8;
  }
  const int grid = g.total < ncu ? g.total : ncu;
  g.stagger_cycles = 0; g.ncu = ncu; g.dbg = 0;
#ifdef CIPS_TUNING
  // stagger quantum = a quarter of one tile's main-loop time (~3500 cycles per k-tile); measured: no effect
  g.stagger_cycles = (cips_tune_env("CIPS_X3_STAGGER", 0) == 1 && g.total >= 2 * grid) ? (d->K / BK) * 3500 / 4 : 0;
  g.dbg = cips_tune_env("CIPS_X3_GDBG", 0);
#endif
  hipLaunchKernelGGL(gemm_bf16x3_kernel<4>, dim3(grid), dim3(512), Cfg<4>::SMEM_BYTES, (hipStream_t)stream, g);
  return CIPS_CHECK_LAUNCH();
}


extern "C" int cips_gemm_bf16x3_fuses_torgb(const cips_gemm_x3_desc* d) {
  if (!d || !d->torgb_w) return 0;
  if (!(x3_wants_256(d) && d->kernel != 3)) return 0;
  return cips_gemm_bf16x3_v3_accepts(d) == 0 ? 1 : 0;
}

extern "C" int cips_gemm_bf16x3_takes_addp(const cips_gemm_x3_desc* d) {
  if (!d || !d->addp_hi) return 0;
  if (!(x3_wants_256(d) && d->kernel != 3)) return 0;
  return cips_gemm_bf16x3_v3_accepts(d) == 0 ? 1 : 0;
}

extern "C" int cips_gemm_bf16x3_km(const cips_gemm_x3_desc* d, cips_stream_t stream) {
  if (!d || d->M <= 0 || d->N <= 0 || d->K <= 0 || d->batch <= 0 || !d->C) return (int)hipErrorInvalidValue;
  if ((d->K & 31) || (d->M & 7) || (d->N & 7) || (d->lda & 7) || (d->ldb & 7) || (d->strideA & 7) || (d->strideB & 7))
    return (int)hipErrorInvalidValue;
  if (d->P_hi || d->T_hi || d->mask || d->add || d->addp_hi || d->rgb_g || d->C_unmasked || d->mask_out || d->res_hi || d->act)
    return (int)hipErrorNotSupported;            // the K-major form has the plain fp32 epilogue only
  // square-ish outputs filling the chip with 256x256 tiles: the wide kernel (a single problem is a group of one);
  // descriptor field `kernel`: 1 never, 2 / 3 whenever the shape allows
  if (d->kernel < 0 || d->kernel > 3) return (int)hipErrorInvalidValue;
  if (d->kernel != 1 && d->M >= 256 && d->N >= 256 &&
      ((long long)((d->M + 255) / 256) * ((d->N + 255) / 256) * d->batch >= 192 || d->kernel >= 2)) {
    const int rc = cips_gemm_bf16x3_km_grouped(d, 1, stream);
    if (rc != (int)hipErrorNotSupported) return rc;
  }
  // 256-row tiles when M fills them, else the 128-row form (SIREN weight gradients: M = 128 / 64)
  const int bm = (d->M > 128) ? 256 : 128;
  Args g;
  g.d = *d;
  g.tiles_m = (d->M + bm - 1) / bm;
  g.tiles_n = (d->N + BN - 1) / BN;
  long long total = (long long)g.tiles_m * g.tiles_n * d->batch;
  if (total > 0x7fffffffLL) return (int)hipErrorInvalidValue;
  g.total = (int)total;
  g.stagger_cycles = 0; g.ncu = 0; g.dbg = 0;
  static int ncu = 0;
  CIPS_PER_DEVICE(ncu, 0);
  if (!ncu) {
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
    if (ncu <= 0) ncu = 256;
    ncu = (ncu / 8) * 8;
    hipFuncSetAttribute((const void*)gemm_bf16x3_km_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, KmCfg<4>::SMEM_BYTES);
    hipFuncSetAttribute((const void*)gemm_bf16x3_km_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, KmCfg<2>::SMEM_BYTES);
  }
  if (bm == 256) {
    const int grid = g.total < ncu ? g.total : ncu;
    hipLaunchKernelGGL(gemm_bf16x3_km_kernel<4>, dim3(grid), dim3(512), KmCfg<4>::SMEM_BYTES, (hipStream_t)stream, g);
  } else {
    const int grid = g.total < 2 * ncu ? g.total : 2 * ncu;
    hipLaunchKernelGGL(gemm_bf16x3_km_kernel<2>, dim3(grid), dim3(256), KmCfg<2>::SMEM_BYTES, (hipStream_t)stream, g);
  }
  return CIPS_CHECK_LAUNCH();
}

extern "C" int cips_split_planes(const float* x, void* p_hi, void* p_lo, void* t_hi, void* t_lo, int rows,
                                 int cols, int ldx, int ldp, int ldt, int batch, long long stride_x,
                                 long long stride_p, long long stride_t, cips_stream_t stream) {
  if (rows <= 0 || cols <= 0 || batch <= 0) return (int)hipErrorInvalidValue;
  dim3 grid((cols + 31) / 32, (rows + 31) / 32, batch);
  hipLaunchKernelGGL(split_planes_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, (u16*)p_hi, (u16*)p_lo,
                     (u16*)t_hi, (u16*)t_lo, rows, cols, ldx, ldp, ldt, stride_x, stride_p, stride_t);
  return CIPS_CHECK_LAUNCH();
}

extern "C" int cips_split_planes_nhwc(const float* x, void* t_hi, void* t_lo, int B, int C, int n, cips_stream_t stream) {
  if (!x || !t_hi || !t_lo || B <= 0 || C <= 0 || n <= 0 || (C & 7)) return (int)hipErrorInvalidValue;
  dim3 grid((n + 63) / 64, (C + 63) / 64, B);
  hipLaunchKernelGGL(split_nhwc_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, (u16*)t_hi, (u16*)t_lo, C, n, B);
  return CIPS_CHECK_LAUNCH();
}

extern "C" int cips_lrelu_bwd_bias_nhwc_tiles(int HW) { return HW > 0 ? (HW + 63) / 64 : 0; }
extern "C" int cips_lrelu_bwd_bias_nhwc(const float* grad, const float* refer, void* t_hi, void* t_lo, float* part, int B, int C,
                                        int HW, float alpha, float scale, cips_stream_t stream) {
  if (!grad || !refer || !t_hi || !t_lo || !part || B <= 0 || C <= 0 || HW <= 0 || (C & 7)) return (int)hipErrorInvalidValue;
  if (B > 65535 || (C + 63) / 64 > 65535) return (int)hipErrorInvalidValue;
  const int ntiles = (HW + 63) / 64;
  dim3 grid((unsigned)ntiles, (unsigned)((C + 63) / 64), (unsigned)B);
  hipLaunchKernelGGL(lrelu_bwd_nhwc_kernel, grid, dim3(256), 0, (hipStream_t)stream, grad, refer, (u16*)t_hi, (u16*)t_lo, part, C, HW, B,
                     ntiles, alpha, scale);
  return CIPS_CHECK_LAUNCH();
}

extern "C" int cips_conv_wgrad_finish(const float* part, float* dw, int nchunks, int taps, int O, int C, float scale,
                                      cips_stream_t stream) {
  if (!part || !dw || nchunks <= 0 || taps <= 0 || O <= 0 || C <= 0) return (int)hipErrorInvalidValue;
  const long long oc = (long long)O * C, total = oc * taps;
  const long long blocks = (total + 255) / 256;
  hipLaunchKernelGGL(conv_wgrad_finish_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, (hipStream_t)stream,
                     part, dw, nchunks, taps, oc, scale);
  return CIPS_CHECK_LAUNCH();
}
