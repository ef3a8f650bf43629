// gemm_bf16x3_v5.hip — EXPERIMENT (round 5): the v3 kernel (same tile, LDS image, LDS-DMA staging, MFMA order per accumulator
// and epilogue: results bit-identical) with a ROLE-SPLIT main loop instead of v3's in-wave interleave.
//
// scripts/probe/hot_operands_probe.py: v3's main loop takes 148 us on zero operands (no power limit), 159 us on L2-resident
// and 169 us on the C2 operands — at most 0.58 of the matrix pipe's cycles even when neither power nor memory limits it.  The
// two waves of a SIMD run the same instruction stream in lockstep behind the per-k-tile barrier, so whenever one of them
// cannot issue an MFMA (fragment-read waits, the eight LDS-DMA instructions per k-tile, the barrier itself) neither can the
// other.  Here the workgroup's waves form two groups, waves 0-3 and 4-7 — one wave of each per SIMD (waves go to SIMDs in
// the cyclic order 0, 2, 1, 3) — that run ONE barrier interval apart: a k-step is  LOAD (12 ds_read_b128, and the LDS-DMA of
// the k-tile after next when its stage has just been freed) | barrier | 24 MFMAs at raised priority | barrier,  and while
// one group is in its MFMA phase the other is in its LOAD phase (MI355X_MICROARCH.md "two waves per SIMD", T3 + T4 + T5 of
// the programming guide).  One fragment register set instead of two.
#include "common.h"
#include "../../include/cips3d_hip.h"
#include <utility>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 256, BN = 256, BK = 32, ROWB = 64;
constexpr int OFF_AHI = 0, OFF_ALO = BM * ROWB, OFF_BHI = 2 * BM * ROWB, OFF_BLO = OFF_BHI + BN * ROWB;
constexpr int STAGE = OFF_BLO + BN * ROWB;      // 65536
constexpr int SCR_OFF = 2 * STAGE;              // the epilogue scratch starts behind the two stages
constexpr int SCR_WAVE = 4096;                  // per wave: two fp32 [16][32] halves (ping-pong)
constexpr int SMEM_BYTES = SCR_OFF + 8 * SCR_WAVE;
static_assert(SMEM_BYTES == 163840, "the kernel owns the whole LDS of the CU");

struct VArgs {
  cips_gemm_x3_desc d;
  int tiles_m, tiles_n, total;
  // tuning aids (env CIPS_X3_V3DBG / CIPS_X3_V3SKEW / CIPS_X3_V3PHASES; results are WRONG with dbg != 0):
  // dbg bit 0: epilogue arithmetic without any global store; bit 1: every store of a workgroup lands in one 256 KiB
  // window (L2-resident); bit 2: non-temporal stores; bit 3: skip the epilogue
  int dbg;
  int skew, phases;     // workgroups of XCD x start (x % phases) * skew shader cycles late
};

template <typename F, int... I>
__device__ __forceinline__ void static_for(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }

// fragment q of a k-step, in the order the pass-major MFMA stream first needs them:
//   0: a_lo[0]   1..4: b_hi[0..3]   5: a_lo[1]   6: a_hi[0]   7..10: b_lo[0..3]   11: a_hi[1]
__device__ __forceinline__ constexpr int frag_is_a(int q) { return q == 0 || q == 5 || q == 6 || q == 11; }
__device__ __forceinline__ constexpr int frag_off(int q) {
  return q == 0 ? OFF_ALO : q == 5 ? OFF_ALO + 32 * ROWB : q == 6 ? OFF_AHI : q == 11 ? OFF_AHI + 32 * ROWB
       : q <= 4 ? OFF_BHI + (q - 1) * 32 * ROWB : OFF_BLO + (q - 7) * 32 * ROWB;
}
__device__ __forceinline__ constexpr int mfma_a(int m) { return (m >> 3) == 0 ? (((m >> 2) & 1) ? 5 : 0) : (((m >> 2) & 1) ? 11 : 6); }
__device__ __forceinline__ constexpr int mfma_b(int m) { return (m >> 3) == 1 ? 7 + (m & 3) : 1 + (m & 3); }

#define LDS_B128(a) (*((__attribute__((address_space(3))) const bf16x8*)(uintptr_t)(a)))
#define LDS_F4(a) (*((__attribute__((address_space(3))) const f32x4*)(uintptr_t)(a)))
#define LDS_W32(a) (*((__attribute__((address_space(3))) float*)(uintptr_t)(a)))
#define SB() __builtin_amdgcn_sched_barrier(0)

// FAST: the epilogue shapes of the training step, decided at compile time — planes out, no fp32 C; forward flavours
// (no gate input) apply the LeakyReLU and write the gate bit plane, backward flavours (gate input) do neither; only
// C_unmasked (the skip gradient of the add flavour) stays a run-time switch.  !FAST: every switch of the descriptor.
// RGBF: ToRGB forward folded in (descriptor fields torgb_w / torgb_part): per row and 128-column block the three partial
// dot products of the final values with the ToRGB weights, accumulated over the wave's four column sub-blocks in
// registers, reduced over the four lanes of a row with two DPP adds and stored as one float4 per row.
// ADDP: the addend (HAS_ADD) arrives as the split planes of a gated tensor plus the bit plane of that gate (descriptor fields
// addp_*): value = (hi + lo) * (bit ? 1 : addp_gain), same bytes in as the fp32 addend, and no C_unmasked copy is needed by
// the next layer.
template <bool HAS_ADD, bool HAS_MASK, bool HAS_RES, bool FAST, bool DBG = false, bool RGBF = false, bool ADDP = false>
__global__ __launch_bounds__(512) void gemm_bf16x3_v5_kernel(VArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const cips_gemm_x3_desc& d = g.d;
  const int tid = threadIdx.x;
  const int lane0 = tid & 63;
  const int uw = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = uw >> 1, wn = uw & 1;                     // 4 x 2 waves, 64 x 128 outputs each
  const int nk = d.K / BK;                                 // even, >= 2
  const unsigned sbase = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)smem);

  auto decode = [&](int t, int& tm, int& tn, int& bz) {    // XCD-contiguous tile ranges (gemm_bf16x3.hip)
    const int nx = 8;
    const int q = g.total / nx, r = g.total % nx;
    const int xcd = t % nx, idx = t / nx;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int bid = base + idx;
    tn = bid % g.tiles_n;
    tm = (bid / g.tiles_n) % g.tiles_m;
    bz = bid / (g.tiles_n * g.tiles_m);
  };
  // LDS-DMA source of one output tile: four uniform plane pointers + one 32-bit byte offset per lane, operand and
  // row-group half (lane L = row L>>2 of the wave's 16-row group, 16-byte slot (L&3) ^ ((row>>2)&3): the swizzle
  // lives in the source address, the LDS image is lane-linear)
  struct Src { const u16 *Ahi, *Alo, *Bhi, *Blo; unsigned offA[2], offB[2]; };
  auto make_src = [&](int tm, int tn, int bz, int lane, Src& sr) {
    sr.Ahi = (const u16*)d.A_hi + (long long)bz * d.strideA + (long long)tm * BM * d.lda;
    sr.Alo = (const u16*)d.A_lo + (long long)bz * d.strideA + (long long)tm * BM * d.lda;
    sr.Bhi = (const u16*)d.B_hi + (long long)bz * d.strideB + (long long)tn * BN * d.ldb;
    sr.Blo = (const u16*)d.B_lo + (long long)bz * d.strideB + (long long)tn * BN * d.ldb;
    const int drow = lane >> 2, dslot = lane & 3;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int row = (uw + 8 * p) * 16 + drow;
      const int kcsw = dslot ^ ((row >> 2) & 3);
      sr.offA[p] = (unsigned)(row * d.lda + kcsw * 8) * 2u;
      sr.offB[p] = (unsigned)(row * d.ldb + kcsw * 8) * 2u;
    }
  };
  auto dma = [&](const u16* plane_k, unsigned off, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(off), "s"(plane_k), "s"(lds_addr) : "memory");
  };
  // piece pc = 0..7 of the k-tile starting at contraction index k0, into the stage at LDS byte offset `st`
  auto dma_piece = [&](const Src& sr, int pc, int k0, unsigned st) {
    const int pp = pc >> 2, which = pc & 3;
    const unsigned la = sbase + st + (unsigned)((uw + 8 * pp) * 16 * ROWB);
    if (which == 0) dma(sr.Ahi + k0, sr.offA[pp], la + OFF_AHI);
    else if (which == 1) dma(sr.Alo + k0, sr.offA[pp], la + OFF_ALO);
    else if (which == 2) dma(sr.Bhi + k0, sr.offB[pp], la + OFF_BHI);
    else dma(sr.Blo + k0, sr.offB[pp], la + OFF_BLO);
  };

  if (CIPS_TUNE(g.skew) > 0) {
    const int ph = (int)(blockIdx.x & 7) % (g.phases > 0 ? g.phases : 1);
    const long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < (long long)ph * g.skew) __builtin_amdgcn_s_sleep(32);
  }
  // ---- kernel prologue: the first tile's k-tiles 0 and 1
  bool have = (int)blockIdx.x < g.total;
  if (have) {
    int tm, tn, bz;
    decode(blockIdx.x, tm, tn, bz);
    Src s0;
    make_src(tm, tn, bz, lane0, s0);
#pragma unroll
    for (int pc = 0; pc < 8; ++pc) dma_piece(s0, pc, 0, 0);
#pragma unroll
    for (int pc = 0; pc < 8; ++pc) dma_piece(s0, pc, BK, STAGE);
  }
  int younger = 0;        // lower bound of the VMEM operations issued after the 16 DMA pieces of the coming tile

  for (int tseq = blockIdx.x; tseq < g.total; tseq += gridDim.x) {
    int lane = lane0;
    asm volatile("" : "+v"(lane));        // keeps the per-lane epilogue addresses out of the persistent loop's preheader
    const int l31 = lane & 31, hf = lane >> 5;
    int tm, tn, bz;
    decode(tseq, tm, tn, bz);
    const int m0 = tm * BM, n0 = tn * BN;
    Src src;
    make_src(tm, tn, bz, lane, src);
    const int tnext = tseq + (int)gridDim.x;
    const bool have_next = tnext < g.total;

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment read addresses: row R = tile row + lane&31, 16-byte chunk (2 ks + hf) ^ ((R>>2)&3); one lane base per
    // operand and k-step, everything else is an immediate
    const int csw = (l31 >> 2) & 3;
    const unsigned fa0 = sbase + (wm * 64 + l31) * ROWB + ((hf ^ csw) << 4), fa1 = sbase + (wm * 64 + l31) * ROWB + (((2 + hf) ^ csw) << 4);
    const unsigned fb0 = sbase + (wn * 128 + l31) * ROWB + ((hf ^ csw) << 4), fb1 = sbase + (wn * 128 + l31) * ROWB + (((2 + hf) ^ csw) << 4);

    bf16x8 F[12];

    // ---- epilogue plumbing declared here: the first inputs are requested inside the last k-tile
    const bool win = DBG && (g.dbg & 2);             // tuning: every tile's outputs land in rows 0..255 of image 0
    const long long cbase = win ? n0 : (long long)bz * d.strideC + (long long)m0 * d.ldc + n0;      // fp32 tensors (ld = ldc)
    const long long pbase = win ? n0 : (long long)bz * d.strideP + (long long)m0 * d.ldp + n0;      // planes / gate planes (ld = ldp)
    auto st16 = [&](void* ubase, unsigned off, u32x4 v) {
      u32x4* q = (u32x4*)((char*)ubase + off);
      if constexpr (DBG) {
        if (g.dbg & 1) return;
        if (g.dbg & 4) { __builtin_nontemporal_store(v, q); return; }
      }
      *q = v;
    };
    auto st16f = [&](void* ubase, unsigned off, float a, float b, float c, float e) {
      u32x4 v = {__float_as_uint(a), __float_as_uint(b), __float_as_uint(c), __float_as_uint(e)};
      st16(ubase, off, v);
    };
    const int h_rr = lane >> 2, q2 = lane & 3;
    const unsigned eC = (unsigned)((wm * 64 + h_rr) * d.ldc + wn * 128 + q2 * 8);         // lane element offsets in the tile
    const unsigned eP = (unsigned)((wm * 64 + h_rr) * d.ldp + wn * 128 + q2 * 8);
    // half-sub-tile hs = 0..15: column block jj = hs>>2, row block si = (hs>>1)&1, half h = hs&1 (16 rows)
    // (RGBF: the other nesting — row set (si, h) = hs>>2 outermost, column block jj = hs&3 innermost — so that a row's ToRGB
    // sums are complete after four consecutive steps and only one set of three accumulators is live)
    // half-sub-tile hs = (row set hs >> 2, column block hs & 3): the column block runs innermost, so the four 64-byte
    // pieces of a row's 256 output bytes are stored back to back and leave L2 as whole lines (measured against the
    // column-block-outermost order on the C2 shape: plain 232 -> 210 us, res 280 -> 266, gate 236 -> 220, add 363 -> 351)
    auto JJ = [](int hs) -> int { return hs & 3; };
    auto RS = [](int hs) -> int { return hs >> 2; };
    auto uoffC = [&](int hs) -> long long { return (long long)(((RS(hs) >> 1) * 32 + (RS(hs) & 1) * 16)) * d.ldc + JJ(hs) * 32; };
    auto uoffP = [&](int hs) -> long long { return (long long)(((RS(hs) >> 1) * 32 + (RS(hs) & 1) * 16)) * d.ldp + JJ(hs) * 32; };
    struct Pre { float4 add[(HAS_ADD && !ADDP) ? 2 : 1]; float gg[HAS_ADD ? 3 : 1]; unsigned mask, amask; uint4 rh, rl; };      // rh / rl: residual planes (HAS_RES) or the planes addend (ADDP)
    const bool has_rgb = HAS_ADD && d.rgb_g != nullptr;
    auto prefetch = [&](int hs, Pre& p) {
      if constexpr (HAS_ADD) {
        if constexpr (ADDP) {
          const char* qh = (const char*)((const u16*)d.addp_hi + pbase + uoffP(hs));
          const char* ql = (const char*)((const u16*)d.addp_lo + pbase + uoffP(hs));
          p.rh = *reinterpret_cast<const uint4*>(qh + eP * 2u);
          p.rl = *reinterpret_cast<const uint4*>(ql + eP * 2u);
          const unsigned char* q = (const unsigned char*)d.addp_gate + ((pbase + uoffP(hs)) >> 3);
          p.amask = q[eP >> 3];
        } else {
          const char* q = (const char*)(d.add + cbase + uoffC(hs));
          p.add[0] = *reinterpret_cast<const float4*>(q + eC * 4u);
          p.add[1] = *reinterpret_cast<const float4*>(q + eC * 4u + 16);
        }
        if (has_rgb) {
          const int row = m0 + wm * 64 + (RS(hs) >> 1) * 32 + (RS(hs) & 1) * 16 + h_rr;
          const float* gp = d.rgb_g + ((long long)bz * d.M + row) * 3;
          p.gg[0] = gp[0]; p.gg[1] = gp[1]; p.gg[2] = gp[2];
        }
      }
      if constexpr (HAS_MASK) {
        const unsigned char* q = (const unsigned char*)d.mask + ((pbase + uoffP(hs)) >> 3);
        p.mask = q[eP >> 3];
      }
      if constexpr (HAS_RES) {
        const char* qh = (const char*)((const u16*)d.res_hi + pbase + uoffP(hs));
        const char* ql = (const char*)((const u16*)d.res_lo + pbase + uoffP(hs));
        p.rh = *reinterpret_cast<const uint4*>(qh + eP * 2u);
        p.rl = *reinterpret_cast<const uint4*>(ql + eP * 2u);
      }
    };
    constexpr bool HAS_IN = HAS_ADD || HAS_MASK || HAS_RES;
    constexpr int NPF = (HAS_ADD || RGBF) ? 3 : 4;        // ring of epilogue-input slots; slot hs % NPF is refilled NPF-1 half-sub-tiles ahead
    Pre pre[NPF];

    // ---- tile start: k-tile 0 has landed everywhere
    if (younger >= 24) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");      // 8 pieces of k-tile 1 + >= 24 younger ops stay in flight
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    // the stagger: group 1 enters the loop one barrier late (and group 0 leaves it one barrier late).  Barrier #j of one
    // group is barrier #j of the other; between #(2s+1) and #(2s+2) group 0 runs MFMA(s) and group 1 LOAD(s), between
    // #(2s+2) and #(2s+3) group 0 runs LOAD(s+1) and group 1 MFMA(s).
    const int grp = uw >> 2;
    if (grp == 1) asm volatile("s_barrier" ::: "memory");
    const bool young = younger >= 24;
    // wait for the LDS-DMA of the NEXT k-tile before the barrier that precedes its first read (k-step s odd): after k-step 1
    // that is k-tile 1 (issued before the previous epilogue, whose >= 24 stores may stay in flight), later the k-tile this
    // wave requested two k-steps ago
    auto dma_wait = [&](int s_) {
      if (s_ == 1 && young) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    for (int s_ = 0; s_ < 2 * nk; ++s_) {
      const unsigned st = (unsigned)((s_ >> 1) & 1) * STAGE;
      // ---- LOAD(s): the twelve fragments of k-step s; on the first k-step of k-tile t >= 1 also the LDS-DMA of k-tile t + 1
      // into the other stage (k-tile t - 1's: every wave's last read of it lies behind a barrier this wave has passed)
      {
        unsigned a_ = ((s_ & 1) ? fa1 : fa0) + st, b_ = ((s_ & 1) ? fb1 : fb0) + st;
        asm volatile("" : "+v"(a_), "+v"(b_));
#pragma unroll
        for (int q = 0; q < 12; ++q) F[q] = LDS_B128((frag_is_a(q) ? a_ : b_) + frag_off(q));
        SB();
        if ((s_ & 1) == 0 && s_ >= 2 && (s_ >> 1) + 1 < nk) {
#pragma unroll
          for (int pc = 0; pc < 8; ++pc) dma_piece(src, pc, ((s_ >> 1) + 1) * BK, STAGE - st);
        }
        if (grp == 1 && (s_ & 1)) dma_wait(s_);
      }
      asm volatile("s_barrier" ::: "memory");
      // ---- MFMA(s)
      __builtin_amdgcn_s_setprio(1);
      static_for(std::make_integer_sequence<int, 24>{}, [&](auto M_) {
        constexpr int m = decltype(M_)::value;
        acc[(m >> 2) & 1][m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[mfma_a(m)], F[mfma_b(m)], acc[(m >> 2) & 1][m & 3], 0, 0, 0);
      });
      __builtin_amdgcn_s_setprio(0);
      SB();
      if (grp == 0 && (s_ & 1)) dma_wait(s_);
      asm volatile("s_barrier" ::: "memory");
    }
    if (grp == 0) asm volatile("s_barrier" ::: "memory");
    // both stages are free: the next output tile's first two k-tiles land under the epilogue
    if (have_next) {
      int tm2, tn2, bz2;
      decode(tnext, tm2, tn2, bz2);
      Src nsrc;
      make_src(tm2, tn2, bz2, lane, nsrc);
#pragma unroll
      for (int pc = 0; pc < 8; ++pc) dma_piece(nsrc, pc, 0, 0);
#pragma unroll
      for (int pc = 0; pc < 8; ++pc) dma_piece(nsrc, pc, BK, STAGE);
    }
    if constexpr (HAS_IN) {
#pragma unroll
      for (int i = 0; i < NPF - 1; ++i) prefetch(i, pre[i]);
    }

    // ---- epilogue (order of operations: +add, +rgb term, C_unmasked, gate, act, mask_out, +res, outputs)
    u16* Phi = (u16*)d.P_hi; u16* Plo = (u16*)d.P_lo;
    const bool do_act = FAST ? !HAS_MASK : (d.act != 0);
    const bool do_bits = FAST ? !HAS_MASK : (d.mask_out != nullptr);
    const bool do_c = FAST ? false : (d.C != nullptr);
    const bool do_p = FAST ? true : (Phi != nullptr);
    const bool do_cu = (FAST && !HAS_ADD) ? false : (d.C_unmasked != nullptr);
    younger = (has_rgb || !(do_p || do_c)) ? 0 : 24;
    const unsigned sw = sbase + SCR_OFF + uw * SCR_WAVE;
    // scratch image of a half: fp32 [16][32], the 16-byte chunk index of row R XORed with (R>>2)&1: the MFMA layout's
    // writes (lane = column) and the row-contiguous 16-byte reads are both conflict-free
    const unsigned wb = sw + (4 * hf) * 128 + ((((l31 >> 2) ^ hf)) << 4) + (l31 & 3) * 4;
    const unsigned rsw = (h_rr >> 2) & 1;
    const unsigned rb0 = sw + h_rr * 128 + (((2 * q2) ^ rsw) << 4), rb1 = sw + h_rr * 128 + (((2 * q2 + 1) ^ rsw) << 4);
    auto put = [&](auto HS_) {                // accumulators of half-sub-tile hs -> scratch half hs & 1
      constexpr int hs = decltype(HS_)::value, jj = hs & 3, rsx = hs >> 2, si = rsx >> 1, h = rsx & 1;
#pragma unroll
      for (int r = 0; r < 8; ++r)
        LDS_W32(wb + (hs & 1) * 2048 + ((r & 3) + 8 * (r >> 2)) * 128) = acc[si][jj][8 * h + r];
    };
    float rw[3][8];                           // rgb_w columns of the current column block
    float tacc[3] = {0.f, 0.f, 0.f};          // RGBF: the current row set's ToRGB sums
    if (DBG && (g.dbg & 8)) { asm volatile("" :: "v"(acc[0][0][0]), "v"(acc[1][3][15])); continue; }
    put(std::integral_constant<int, 0>{});
    static_for(std::make_integer_sequence<int, 16>{}, [&](auto HS_) {
      constexpr int hs = decltype(HS_)::value;
      if constexpr (hs + 1 < 16) put(std::integral_constant<int, hs + 1>{});
      float y[8];
      {
        const f32x4 a = LDS_F4(rb0 + (hs & 1) * 2048), b = LDS_F4(rb1 + (hs & 1) * 2048);
        y[0] = a[0]; y[1] = a[1]; y[2] = a[2]; y[3] = a[3]; y[4] = b[0]; y[5] = b[1]; y[6] = b[2]; y[7] = b[3];
      }
      Pre& cur = pre[hs % NPF];
      const long long uc = uoffC(hs), up = uoffP(hs);
      if constexpr (HAS_ADD) {
        if constexpr (ADDP) {
          const unsigned wh[4] = {cur.rh.x, cur.rh.y, cur.rh.z, cur.rh.w}, wl[4] = {cur.rl.x, cur.rl.y, cur.rl.z, cur.rl.w};
          const unsigned am = cur.amask;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v0 = __uint_as_float(wh[e] << 16) + __uint_as_float(wl[e] << 16);
            const float v1 = __uint_as_float(wh[e] & 0xffff0000u) + __uint_as_float(wl[e] & 0xffff0000u);
            const int s0 = ((int)(am << (31 - 2 * e))) >> 31, s1 = ((int)(am << (30 - 2 * e))) >> 31;
            const float t0 = v0 * d.addp_gain, t1 = v1 * d.addp_gain;
            y[2 * e] += __int_as_float((s0 & __float_as_int(v0)) | (~s0 & __float_as_int(t0)));
            y[2 * e + 1] += __int_as_float((s1 & __float_as_int(v1)) | (~s1 & __float_as_int(t1)));
          }
        } else {
          y[0] += cur.add[0].x; y[1] += cur.add[0].y; y[2] += cur.add[0].z; y[3] += cur.add[0].w;
          y[4] += cur.add[1].x; y[5] += cur.add[1].y; y[6] += cur.add[1].z; y[7] += cur.add[1].w;
        }
        if (has_rgb) {                                       // rank-3 term: + g[row][0..2] . rgb_w[0..2][col..col+8]
          {
            const int col = n0 + wn * 128 + (hs & 3) * 32 + q2 * 8;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              const float4 w0 = *reinterpret_cast<const float4*>(d.rgb_w + (long long)c * d.N + col);
              const float4 w1 = *reinterpret_cast<const float4*>(d.rgb_w + (long long)c * d.N + col + 4);
              rw[c][0] = w0.x; rw[c][1] = w0.y; rw[c][2] = w0.z; rw[c][3] = w0.w; rw[c][4] = w1.x; rw[c][5] = w1.y; rw[c][6] = w1.z; rw[c][7] = w1.w;
            }
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) y[e] = fmaf(cur.gg[0], rw[0][e], fmaf(cur.gg[1], rw[1][e], fmaf(cur.gg[2], rw[2][e], y[e])));
        }
      }
      if (do_cu) {
        float* q = d.C_unmasked + cbase + uc;
        st16f(q, eC * 4u, y[0], y[1], y[2], y[3]);
        st16f(q, eC * 4u + 16, y[4], y[5], y[6], y[7]);
      }
      if constexpr (HAS_MASK) {                              // y *= gate ? 1 : slope, as a bit select
        const unsigned w = cur.mask;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int sel = ((int)(w << (31 - e))) >> 31;
          const float t = y[e] * d.slope;
          y[e] = __int_as_float((sel & __float_as_int(y[e])) | (~sel & __float_as_int(t)));
        }
      }
      unsigned bits = 0;
      if (do_act) {
        if (do_bits) {
          // LeakyReLU and the gate bit from one compare: vcc = y > 0; y = vcc ? y : slope*y; bits = 2*bits + vcc
#pragma unroll
          for (int e = 7; e >= 0; --e) {
            const float t = y[e] * d.slope;
            float o;
            asm volatile("v_cmp_lt_f32 vcc, 0, %2\n\tv_cndmask_b32 %1, %3, %2, vcc\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
                         : "+v"(bits), "=&v"(o) : "v"(y[e]), "v"(t) : "vcc");
            y[e] = o;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) y[e] = lrelu(y[e], d.slope);
        }
      } else if (do_bits) {
#pragma unroll
        for (int e = 0; e < 8; ++e) bits |= (y[e] > 0.f ? 1u : 0u) << e;
      }
      if (do_bits) {
        // the quad's four bytes as one dword, stored by its first lane
        unsigned v = bits << (8 * q2);
        v |= (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);     // quad_perm [1,0,3,2]
        v |= (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);     // quad_perm [2,3,0,1]
        if (q2 == 0 && !(DBG && (g.dbg & 1))) {
          unsigned char* q = (unsigned char*)d.mask_out + ((pbase + up) >> 3);
          *reinterpret_cast<unsigned*>(q + (eP >> 3)) = v;
        }
      }
      if constexpr (HAS_RES) {
        const unsigned wh[4] = {cur.rh.x, cur.rh.y, cur.rh.z, cur.rh.w}, wl[4] = {cur.rl.x, cur.rl.y, cur.rl.z, cur.rl.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          y[2 * e] += __uint_as_float(wh[e] << 16) + __uint_as_float(wl[e] << 16);
          y[2 * e + 1] += __uint_as_float(wh[e] & 0xffff0000u) + __uint_as_float(wl[e] & 0xffff0000u);
        }
      }
      if constexpr (RGBF) {
        constexpr int jj = hs & 3, rs = hs >> 2;             // column sub-block (innermost), row set (si, h)
        const int col = n0 + wn * 128 + jj * 32 + q2 * 8;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float4 w0 = *reinterpret_cast<const float4*>(d.torgb_w + (long long)c * d.N + col);
          const float4 w1 = *reinterpret_cast<const float4*>(d.torgb_w + (long long)c * d.N + col + 4);
          float t = (jj == 0) ? 0.f : tacc[c];
          t = fmaf(y[0], w0.x, t); t = fmaf(y[1], w0.y, t); t = fmaf(y[2], w0.z, t); t = fmaf(y[3], w0.w, t);
          t = fmaf(y[4], w1.x, t); t = fmaf(y[5], w1.y, t); t = fmaf(y[6], w1.z, t); t = fmaf(y[7], w1.w, t);
          tacc[c] = t;
        }
        if constexpr (jj == 3) {
          float v[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            float t = tacc[c];
            t += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(t), 0xB1, 0xF, 0xF, true));
            t += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(t), 0x4E, 0xF, 0xF, true));
            v[c] = t;
          }
          if (q2 == 0 && !(DBG && (g.dbg & 1))) {
            const long long row = (long long)bz * d.M + m0 + wm * 64 + (rs >> 1) * 32 + (rs & 1) * 16 + h_rr;
            float* q = d.torgb_part + ((long long)(tn * 2 + wn) * d.batch * d.M + row) * 4;
            *reinterpret_cast<float4*>(q) = make_float4(v[0], v[1], v[2], 0.f);
          }
        }
      }
      if (do_c) {
        float* q = d.C + cbase + uc;
        st16f(q, eC * 4u, y[0], y[1], y[2], y[3]);
        st16f(q, eC * 4u + 16, y[4], y[5], y[6], y[7]);
      }
      if (do_p) {
        unsigned hw[4], lw[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const f32x2 t = {y[2 * e], y[2 * e + 1]};
          hw[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
          const f32x2 l = {y[2 * e] - __uint_as_float(hw[e] << 16), y[2 * e + 1] - __uint_as_float(hw[e] & 0xffff0000u)};
          lw[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(l, bf16x2));
        }
        const u32x4 vh = {hw[0], hw[1], hw[2], hw[3]}, vl = {lw[0], lw[1], lw[2], lw[3]};
        st16(Phi + pbase + up, eP * 2u, vh);
        st16(Plo + pbase + up, eP * 2u, vl);
      }
      if constexpr (HAS_IN && hs + NPF - 1 < 16) prefetch(hs + NPF - 1, pre[(hs + NPF - 1) % NPF]);
    });
  }  // persistent tile loop
}

}  // namespace

template <bool A, bool Mk, bool R, bool FAST, bool DBG = false, bool RGBF = false, bool ADDP = false>
static void launch_v5f(const VArgs& g, int grid, hipStream_t stream) {
  static bool attr = false;
  CIPS_PER_DEVICE(attr, false);
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm_bf16x3_v5_kernel<A, Mk, R, FAST, DBG, RGBF, ADDP>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    attr = true;
  }
  hipLaunchKernelGGL((gemm_bf16x3_v5_kernel<A, Mk, R, FAST, DBG, RGBF, ADDP>), dim3(grid), dim3(512), SMEM_BYTES, stream, g);
}
template <bool A, bool Mk, bool R>
static void launch_v5(const VArgs& g, int grid, hipStream_t stream) {
  const cips_gemm_x3_desc& d = g.d;
  // the training step's epilogues take the compile-time form
  const bool fwd = !Mk && d.act == 1 && d.mask_out != nullptr, bwd = Mk && d.act == 0 && d.mask_out == nullptr;
  const bool fast = d.P_hi != nullptr && d.C == nullptr && (fwd || bwd) && (A || d.C_unmasked == nullptr);
  if constexpr (!A && !Mk) {
    if (d.torgb_w) { launch_v5f<A, Mk, R, true, false, true>(g, grid, stream); return; }     // the entry point checked `fast`
  }
  if constexpr (A && Mk && !R) {
    if (d.addp_hi) { launch_v5f<A, Mk, R, true, false, false, true>(g, grid, stream); return; }   // likewise
  }
#ifdef CIPS_TUNING
  if (fast && g.dbg) { launch_v5f<A, Mk, R, true, true>(g, grid, stream); return; }
#endif
  if (fast) launch_v5f<A, Mk, R, true>(g, grid, stream);
  else launch_v5f<A, Mk, R, false>(g, grid, stream);
}

// 0: this kernel takes the descriptor; else the error code cips_gemm_bf16x3_v3 returns for it
static int v5_accepts(const cips_gemm_x3_desc* d) {
  if (!d || d->M <= 0 || d->N <= 0 || d->K <= 0 || d->batch <= 0) return (int)hipErrorInvalidValue;
  if ((d->M % BM) || (d->N % BN) || (d->K % (2 * BK))) return (int)hipErrorNotSupported;
  if ((d->lda & 7) || (d->ldb & 7) || (d->strideA & 7) || (d->strideB & 7)) return (int)hipErrorInvalidValue;
  if (d->T_hi || (d->ldc & 3) || (d->strideC & 3) || (d->ldp & 31) || (d->strideP & 31)) return (int)hipErrorNotSupported;
  const bool a = d->add != nullptr || d->addp_hi != nullptr, m = d->mask != nullptr, r = d->res_hi != nullptr;
  if ((a && !m) || (r && (a || m))) return (int)hipErrorNotSupported;
  if (d->addp_hi) {      // planes addend: the backward flavours' compile-time epilogue only
    if (d->add || !d->addp_lo || !d->addp_gate || !(d->act == 0 && !d->mask_out && d->P_hi && !d->C)) return (int)hipErrorNotSupported;
    if (((uintptr_t)d->addp_hi & 15) || ((uintptr_t)d->addp_lo & 15)) return (int)hipErrorNotSupported;
  }
  if ((m && !(d->gate_bits & 1)) || (d->mask_out && !(d->gate_bits & 2))) return (int)hipErrorNotSupported;   // bit planes only
  if (d->rgb_g && !a) return (int)hipErrorNotSupported;
  if (d->mask_out && ((uintptr_t)d->mask_out & 3)) return (int)hipErrorNotSupported;                         // dword stores of the bit plane
  // 32-bit lane offsets
  if ((long long)BM * d->lda * 2 >= 0x7fffffffLL || (long long)BN * d->ldb * 2 >= 0x7fffffffLL ||
      (long long)BM * d->ldc * 4 >= 0x7fffffffLL || (long long)BM * d->ldp * 2 >= 0x7fffffffLL) return (int)hipErrorNotSupported;
  if (d->torgb_w) {      // fused ToRGB: the forward flavours' compile-time epilogue only
    if (!d->torgb_part || a || m || !(d->act == 1 && d->mask_out && d->P_hi && !d->C && !d->C_unmasked)) return (int)hipErrorNotSupported;
    if ((d->N & 127) || ((uintptr_t)d->torgb_part & 15) || ((uintptr_t)d->torgb_w & 15)) return (int)hipErrorNotSupported;
  }
  return 0;
}
extern "C" CIPS_INTERNAL int cips_gemm_bf16x3_v5_accepts(const cips_gemm_x3_desc* d) { return v5_accepts(d); }

// Internal entry (called by cips_gemm_bf16x3 ahead of the wide kernel): same descriptor.  hipErrorNotSupported for
// every shape / epilogue it has no code for.
extern "C" CIPS_INTERNAL int cips_gemm_bf16x3_v5(const cips_gemm_x3_desc* d, cips_stream_t stream) {
  { const int rc = v5_accepts(d); if (rc) return rc; }
  const bool a = d->add != nullptr || d->addp_hi != nullptr, m = d->mask != nullptr, r = d->res_hi != nullptr;
  VArgs g = {};
  g.d = *d;
  g.tiles_m = d->M / BM;
  g.tiles_n = d->N / BN;
  const long long total = (long long)g.tiles_m * g.tiles_n * d->batch;
  if (total > 0x7fffffffLL) return (int)hipErrorInvalidValue;
  g.total = (int)total;
  static int ncu = 0;
  CIPS_PER_DEVICE(ncu, 0);
  if (!ncu) {
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
    if (ncu <= 0) ncu = 256;
    ncu = (ncu / 8) * 8;
  }
  int grid = g.total < ncu ? g.total : ncu;
#ifdef CIPS_TUNING
  {
    // tuning aids (probe builds only), read on every call: see VArgs
    g.dbg = cips_tune_env("CIPS_X3_V3DBG", 0);
    g.skew = cips_tune_env("CIPS_X3_V3SKEW", 0);
    g.phases = cips_tune_env("CIPS_X3_V3PHASES", 2);
    const int eg = cips_tune_env("CIPS_X3_V3GRID", 0);
    if (eg > 0 && eg < grid) grid = (eg / 8) * 8 > 0 ? (eg / 8) * 8 : grid;
  }
#endif
  hipStream_t st = (hipStream_t)stream;
  if (a) launch_v5<true, true, false>(g, grid, st);
  else if (m) launch_v5<false, true, false>(g, grid, st);
  else if (r) launch_v5<false, false, true>(g, grid, st);
  else launch_v5<false, false, false>(g, grid, st);
  return CIPS_CHECK_LAUNCH();
}
