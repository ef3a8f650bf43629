// gemm_bf16x3_wide.hip — 256x256-tile form of the split-bf16 "NT" GEMM (see gemm_bf16x3.hip for the numerics,
// the LDS-DMA staging and the epilogue contract; this file only changes the tiling and the schedule).
//
// Why: at the CIPS head's shapes (M = 131072 rows, N = K = 512) the 256x128 kernel is bound by the bytes its
// workgroups pull through L2 -> LDS (1.57 GB per GEMM, ~10 TB/s: loads alone 120 us, fragment reads + MFMA alone
// 140 us, together 235 us) plus an epilogue whose global reads sit exposed behind it.  Here
//   * the tile is 256 x 256 on eight waves (4 x 2, 64 x 128 each, 128 accumulator registers of the wave's 256;
//     a four-wave 128 x 128 split needs all 256 AGPRs for accumulators and hipcc then shuttles tiles through
//     arch VGPRs every k-step): 1.05 GB of operand traffic per GEMM (64 fp32-flop per byte instead of 43) and
//     12 fragment reads per 24 MFMAs instead of 8 per 12;
//   * two 64 KiB LDS stages; the DMA of k-tile t+1 is issued right after the barrier that publishes tile t and
//     has the whole of tile t's 48 MFMAs per wave to land;
//   * the three passes of the operand split are issued pass-major over the wave's 8 output tiles, so that
//     consecutive MFMAs never chain on one accumulator;
//   * the epilogue walks the wave's 64 x 128 block in eight 32 x 32 sub-tiles: the raw accumulators make ONE trip
//     through a per-wave fp32 LDS scratch to turn "lane = column" into "lane = 8 consecutive columns of a row",
//     and everything else (addend, gate, activation, residual, bf16 splitting with v_cvt_pk_bf16_f32, all
//     stores) happens in that row-contiguous form straight from / to 16-byte global accesses; the global inputs
//     of sub-tile s+2 (residual planes, gate plane, fp32 addend) are requested before sub-tile s is processed —
//     the first two around the last k-tile's MFMAs — so those reads no longer sit exposed behind each other;
//   * the scratch aliases LDS stage 1 only: the first k-tile of the workgroup's NEXT output tile is DMA'd into
//     stage 0 before the epilogue starts, and the next main loop opens with a COUNTED vmcnt that leaves the
//     epilogue's stores in flight (vmcnt retires in order: the DMA pieces are older than every store).
// Restrictions (else the caller falls back to the 256x128 kernel): no transposed planes (T_hi), 16-byte aligned
// rows of every auxiliary tensor (N, ldc, ldp multiples of 8).
#include "common.h"
#include "../../include/cips3d_hip.h"
#include <stdlib.h>
#include <utility>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

constexpr int BM = 256, BN = 256, BK = 32, ROWB = 64;
constexpr int OFF_AHI = 0, OFF_ALO = BM * ROWB, OFF_BHI = 2 * BM * ROWB, OFF_BLO = OFF_BHI + BN * ROWB;
constexpr int STAGE = OFF_BLO + BN * ROWB;      // 65536
constexpr int NSTAGE = 2;
constexpr int PIECES = 8;                       // LDS-DMA instructions per wave per k-tile
constexpr int PF = 36;                          // scratch pitch: fp32 image [32][36] per wave
constexpr int SCR_WAVE = 32 * PF * 4;           // 4608 B per wave
constexpr int SMEM_BYTES = NSTAGE * STAGE;      // 131072; the epilogue scratch (36 KiB) aliases stage 1 only, so that
                                                // the next tile's first k-tile can stream into stage 0 meanwhile
static_assert(8 * SCR_WAVE <= STAGE, "scratch");
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// compile-time loop: the sub-tile index must be a constant, or acc[][] is indexed dynamically and lands in scratch
template <typename F, int... I>
__device__ __forceinline__ void static_for(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }

// Implicit-GEMM convolution (CONV instantiation): the B operand's rows are output pixels of image `bz`, its
// contraction index is (tap, channel); row r of k-tile (tap, c0) is the 32-channel slice at input pixel
// (oy*stride - pad + ky, ox*stride - pad + kx) of the NHWC planes, or the zero row behind the last image.
struct ConvGeom {
  int C, H, W, kw, stride, pad, Wo;
  int nimg;                    // conv2d_x3_v3_kernel: > 0 = the batch is FOLDED into the pixel dimension (output planes of fewer
                               // than 256 pixels: one GEMM column range over B * nimg pixels instead of B mostly empty tiles);
                               // nimg = pixels per image (a multiple of 8), d.N = B * nimg, d.batch = 1
  long long img_stride;        // H*W*C elements
  long long zero_elem;         // element offset of the zero row from the start of the planes
  const float* bias;           // optional per-output-channel bias (row of the GEMM), added before the activation
  int act;                     // 1: y = leaky_relu(y + bias, slope) * act_scale   (FusedLeakyReLU, fused_act.py:47-86)
  float slope, act_scale;
};
// One homogeneous range of output tiles of conv2d_x3_v3_kernel.  A plain convolution is one range; the data gradient of a
// stride-2 convolution is four (one per parity class of the input pixel: every class is a small stride-1 convolution over
// dy with its own tap window, output plane and filter bank), run by ONE persistent launch.
struct ConvPart {
  int ntiles, tiles_n;         // tiles of the range; tiles per output plane along the pixel dimension
  int kw, pad_y, pad_x, Wo;    // tap window width, top / left padding, output plane width
  int N, K, lda;               // output pixels per image (rows of B; a multiple of 8), contraction length, A row pitch
  int nimg;                    // batch folding for this range (see ConvGeom::nimg): N = B * nimg then
  int ldc;                     // output row pitch
  long long a_off, c_off;      // element offsets of this range's filter bank (A planes) and output block (C)
  long long strideC;           // output elements per image
};
struct WArgs {
  cips_gemm_x3_desc d;
  int tiles_m, tiles_n, total, dbg;
  int ksplit;                  // > 1: the contraction is cut into ksplit ranges of k-tiles (chunk c takes k-tiles
                               // [c*T/ksplit, (c+1)*T/ksplit)); chunk c of batch entry b writes C + (c*batch + b)*strideC
  ConvGeom cv;
  int nparts;                  // conv2d_x3_v3_kernel only: 0 = one range described by d / cv / tiles_n (plain convolution)
  ConvPart part[4];
};

__device__ __forceinline__ u16 f2bf(float v) {
  unsigned u = __float_as_uint(v);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (u16)(u >> 16);
}
__device__ __forceinline__ float bf2f(u16 h) { return __uint_as_float(((unsigned)h) << 16); }

// HAS_ADD / HAS_MASK / HAS_RES: which global inputs the epilogue reads (fp32 addend, gate plane, residual planes);
// compile-time so that only their prefetch registers exist.
template <bool HAS_ADD, bool HAS_MASK, bool HAS_RES, bool CONV = false>
__global__ __launch_bounds__(512) void gemm_bf16x3_wide_kernel(WArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const cips_gemm_x3_desc& d = g.d;
  const int tid = threadIdx.x;
  const int lane0 = tid & 63;
  const int uw = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = uw, wm = wave >> 1, wn = wave & 1;       // 4 x 2 waves
  const int M = d.M, N = d.N, K = d.K;
  const int nk_all = CIPS_TUNE(g.dbg & 4) ? 0 : K / BK;
  const int ksplit = g.ksplit > 1 ? g.ksplit : 1;
  const unsigned sbase = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)smem);
#define LDS_B128(a) (*((__attribute__((address_space(3))) const bf16x8*)(uintptr_t)(a)))

  // tile coordinates of sequence number t (XCD-contiguous tile ranges, see gemm_bf16x3.hip)
  auto decode = [&](int t, int& tm, int& tn, int& bz, int& kc) {
    const int nx = 8;
    int q = g.total / nx, r = g.total % nx;
    int xcd = t % nx, idx = t / nx;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int bid = base + idx;
    tn = bid % g.tiles_n;
    tm = (bid / g.tiles_n) % g.tiles_m;
    const int bk = bid / (g.tiles_n * g.tiles_m);
    kc = bk % ksplit;
    bz = bk / ksplit;
  };
  auto chunk_kt0 = [&](int kc) -> int { return (int)((long long)kc * nk_all / ksplit); };
  // LDS-DMA of one k-tile: a wave instruction moves 16 rows x 64 B of one plane; lane L = (row L>>2, slot L&3) fetches
  // the global 16-byte chunk slot ^ ((row>>2)&3) — the swizzle lives in the source address.  Addresses are a uniform
  // plane pointer (advanced by k0 on the scalar side) plus one 32-bit byte offset per piece and lane; rows past
  // M / N are clamped (their products are never stored).
  struct Src { const u16 *Ahi, *Alo, *Bhi, *Blo; unsigned offA[2], offB[2]; int iy0[2], ix0[2]; unsigned chunk[2], zero_rel; int kbase; };
  auto make_src = [&](int tm, int tn, int bz, int kc, int lane, Src& sr) {
    const int m0 = tm * BM, n0 = tn * BN;
    sr.kbase = chunk_kt0(kc) * BK;                         // first contraction index of this tile's chunk
    sr.Ahi = (const u16*)d.A_hi + (long long)bz * d.strideA + (long long)m0 * d.lda + sr.kbase;
    sr.Alo = (const u16*)d.A_lo + (long long)bz * d.strideA + (long long)m0 * d.lda + sr.kbase;
    if constexpr (CONV) {
      sr.Bhi = (const u16*)d.B_hi + (long long)bz * g.cv.img_stride;
      sr.Blo = (const u16*)d.B_lo + (long long)bz * g.cv.img_stride;
      sr.zero_rel = (unsigned)((g.cv.zero_elem - (long long)bz * g.cv.img_stride) * 2);
    } else {
      sr.Bhi = (const u16*)d.B_hi + (long long)bz * d.strideB + (long long)n0 * d.ldb + sr.kbase;
      sr.Blo = (const u16*)d.B_lo + (long long)bz * d.strideB + (long long)n0 * d.ldb + sr.kbase;
    }
    const int drow = lane >> 2, dslot = lane & 3;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int row = (uw + 8 * p) * 16 + drow;
      const int kcsw = dslot ^ ((row >> 2) & 3);
      const int ra = (row < M - m0) ? row : (M - m0 - 1), rb = (row < N - n0) ? row : (N - n0 - 1);
      sr.offA[p] = (unsigned)(ra * d.lda + kcsw * 8) * 2u;
      if constexpr (CONV) {
        const int pix = n0 + rb, oy = pix / g.cv.Wo, ox = pix - oy * g.cv.Wo;
        sr.iy0[p] = oy * g.cv.stride - g.cv.pad;
        sr.ix0[p] = ox * g.cv.stride - g.cv.pad;
        sr.chunk[p] = (unsigned)kcsw * 16u;
      } else {
        sr.offB[p] = (unsigned)(rb * d.ldb + kcsw * 8) * 2u;
      }
    }
  };
  // B-side addressing of k-tile k0: returns the element offset to add to the (uniform) plane pointers; for the
  // convolution it refreshes the two per-lane byte offsets instead (tap = k0 / C, 32 channels from k0 % C)
  auto prep_b = [&](Src& sr, int k0) -> int {
    if constexpr (CONV) {
      const int kk = k0 + sr.kbase;
      const int tap = kk / g.cv.C, c0 = kk - tap * g.cv.C;
      const int ky = tap / g.cv.kw, kx = tap - ky * g.cv.kw;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int iy = sr.iy0[p] + ky, ix = sr.ix0[p] + kx;
        const bool ok = (unsigned)iy < (unsigned)g.cv.H && (unsigned)ix < (unsigned)g.cv.W;
        sr.offB[p] = (ok ? (unsigned)(((iy * g.cv.W + ix) * g.cv.C + c0) * 2) : sr.zero_rel) + sr.chunk[p];
      }
      return 0;
    } else {
      return k0;
    }
  };
  // Written as asm to get the SGPR-base form (uniform plane pointer in s[..], one 32-bit offset per lane): the builtin
  // takes a flat 64-bit pointer and hipcc then builds 64-bit per-lane addresses with two v_lshl_add_u64 per piece
  // (A/B on one box: main loop 207 -> 197 us, 20 fewer VGPRs).
  // M0 (LDS destination of the wave's 1 KiB) is not used by anything else in this kernel.
  auto dma = [&](const u16* plane_k, unsigned off, unsigned char* lds_base) {
    const unsigned la = sbase + (unsigned)(lds_base - smem);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(off), "s"(plane_k), "s"(la) : "memory");
  };
  auto dma_piece = [&](const Src& sr, int pc, int k0, int kb, unsigned char* s) {      // pc = 0..7; kb = prep_b(sr, k0)
    const int pp = pc >> 2, which = pc & 3;
    const int gidx = uw + 8 * pp;                                               // 16 row groups per plane, two per wave
    if (which == 0) dma(sr.Ahi + k0, sr.offA[pp], s + OFF_AHI + gidx * 16 * ROWB);
    else if (which == 1) dma(sr.Alo + k0, sr.offA[pp], s + OFF_ALO + gidx * 16 * ROWB);
    else if (which == 2) dma(sr.Bhi + kb, sr.offB[pp], s + OFF_BHI + gidx * 16 * ROWB);
    else dma(sr.Blo + kb, sr.offB[pp], s + OFF_BLO + gidx * 16 * ROWB);
  };

  bool first_issued = false;       // k-tile 0 of the coming tile is already in flight (issued before the last epilogue)
  int ops_after = 0;               // VMEM operations issued after those DMA pieces

  for (int tseq = blockIdx.x; tseq < g.total; tseq += gridDim.x) {
    // the whole tile body is invariant across the persistent loop: launder the lane id, or hipcc hoists every
    // per-lane address of the epilogue out of the loop and spills them
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int l31 = lane & 31, hf = lane >> 5;
    int tm, tn, bz, kc;
    decode(tseq, tm, tn, bz, kc);
    const int m0 = tm * BM, n0 = tn * BN;
    const int nk = chunk_kt0(kc + 1) - chunk_kt0(kc);      // k-tiles of this tile's chunk (all of them when ksplit == 1)
    Src src;
    make_src(tm, tn, bz, kc, lane, src);

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Fragment reads: row R = tile row + lane&31, 16-byte chunk (2ks + hf) ^ ((R>>2)&3).  (R>>2)&3 only depends on
    // the lane, so every address is one of four lane bases (A / B, ks = 0 / 1) plus an immediate.
    const int csw = (l31 >> 2) & 3;
    const unsigned fa0 = sbase + (wm * 64 + l31) * ROWB + ((hf ^ csw) << 4), fa1 = sbase + (wm * 64 + l31) * ROWB + (((2 + hf) ^ csw) << 4);
    const unsigned fb0 = sbase + (wn * 128 + l31) * ROWB + ((hf ^ csw) << 4), fb1 = sbase + (wn * 128 + l31) * ROWB + (((2 + hf) ^ csw) << 4);
    // One k-tile = two k-steps of 24 MFMAs (pass-major over the 8 output tiles).  The 8 LDS-DMA pieces of the NEXT
    // k-tile are issued one per three MFMAs inside the first k-step: an LDS-DMA issue costs the wave 60-180
    // cycles, which the matrix pipe spends on the MFMAs already queued instead of idling behind a burst of 8.
    auto compute = [&](int stage, bool issue_next, int next_stage, int k0n) {
      unsigned ba[2], bb[2];
      ba[0] = fa0 + stage * STAGE; ba[1] = fa1 + stage * STAGE; bb[0] = fb0 + stage * STAGE; bb[1] = fb1 + stage * STAGE;
      asm volatile("" : "+v"(ba[0]), "+v"(ba[1]), "+v"(bb[0]), "+v"(bb[1]));   // one base register per operand and k-step,
                                                                                // constants go to the offset field
      unsigned char* sn = smem + next_stage * STAGE;
      const int kbn = issue_next ? prep_b(src, k0n) : 0;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 ah[2], al[2], bh[4], bl[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          ah[i] = LDS_B128(ba[ks] + OFF_AHI + i * 32 * ROWB);
          al[i] = LDS_B128(ba[ks] + OFF_ALO + i * 32 * ROWB);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          bh[j] = LDS_B128(bb[ks] + OFF_BHI + j * 32 * ROWB);
          bl[j] = LDS_B128(bb[ks] + OFF_BLO + j * 32 * ROWB);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 24; ++m) {
          const int pass = m >> 3, i = (m >> 2) & 1, j = m & 3;
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pass == 0 ? al[i] : ah[i], pass == 1 ? bl[j] : bh[j], acc[i][j], 0, 0, 0);
          if (ks == 0 && (m % 3) == 2) {
            __builtin_amdgcn_sched_barrier(0);
            if (issue_next) dma_piece(src, m / 3, k0n, kbn, sn);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };

    // ---------------- epilogue plumbing (declared first: the first sub-tiles' inputs are requested around the last k-tile)
    const long long cb = ((long long)kc * d.batch + bz) * d.strideC, pb = (long long)bz * d.strideP;
    u16* Phi = (u16*)d.P_hi; u16* Plo = (u16*)d.P_lo;
    float* sc_f = reinterpret_cast<float*>(smem + STAGE + wave * SCR_WAVE);      // inside stage 1
    // sub-tile st = 4*i + jj: rows wm*64 + 32 i .., columns wn*128 + 32 jj ..   (one MFMA tile).  Row-contiguous form:
    // lane -> rows h_rr and h_rr + 16, 8 consecutive columns from h_c8
    const int h_rr = lane >> 2, h_c8 = (lane & 3) * 8;
    struct Pre { float4 add[HAS_ADD ? 4 : 1]; uint4 mask[HAS_MASK ? 2 : 1], rh[HAS_RES ? 2 : 1], rl[HAS_RES ? 2 : 1]; };
    auto prefetch = [&](int st, Pre& p) {
      const int row0 = m0 + wm * 64 + (st >> 2) * 32, col0 = n0 + wn * 128 + (st & 3) * 32;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int row = row0 + h_rr + 16 * c, col = col0 + h_c8;
        const bool ok = row < M && col < N;
        if constexpr (HAS_ADD) {
          const float* q = d.add + cb + (long long)row * d.ldc + col;
          p.add[2 * c] = ok ? *reinterpret_cast<const float4*>(q) : make_float4(0.f, 0.f, 0.f, 0.f);
          p.add[2 * c + 1] = ok ? *reinterpret_cast<const float4*>(q + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const long long o = pb + (long long)row * d.ldp + col;
        if constexpr (HAS_MASK) {
          if (d.gate_bits & 1) {       // bit plane: this lane's 8 columns are one byte
            p.mask[c] = make_uint4(ok ? (unsigned)((const unsigned char*)d.mask)[o >> 3] : 0u, 0, 0, 0);
          } else {
            p.mask[c] = ok ? *reinterpret_cast<const uint4*>((const u16*)d.mask + o) : make_uint4(0, 0, 0, 0);
          }
        }
        if constexpr (HAS_RES) {
          p.rh[c] = ok ? *reinterpret_cast<const uint4*>((const u16*)d.res_hi + o) : make_uint4(0, 0, 0, 0);
          p.rl[c] = ok ? *reinterpret_cast<const uint4*>((const u16*)d.res_lo + o) : make_uint4(0, 0, 0, 0);
        }
      }
    };
    constexpr int LOADS_PER_SUB = (HAS_ADD ? 4 : 0) + (HAS_MASK ? 2 : 0) + (HAS_RES ? 4 : 0);

    // ---------------- main loop: two stages, DMA of k-tile kt+1 in flight under the MFMAs of k-tile kt
    Pre pre[2];
    if (nk > 0 && !first_issued && !CIPS_TUNE(g.dbg & 2)) {
      const int kb0 = prep_b(src, 0);
#pragma unroll
      for (int pc = 0; pc < 8; ++pc) dma_piece(src, pc, 0, kb0, smem);
    }
    for (int kt = 0; kt < nk; ++kt) {
      if (kt == 0 && first_issued) {
        // k-tile 0 was issued before the previous epilogue: leave that epilogue's younger stores / loads in flight
        if (ops_after >= 56) asm volatile("s_waitcnt vmcnt(56)" ::: "memory");
        else if (ops_after >= 48) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
        else if (ops_after >= 40) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
        else if (ops_after >= 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        else if (ops_after >= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else if (ops_after >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();              // k-tile kt has landed everywhere; stage (kt+1)&1 is free again
      const bool more = kt + 1 < nk;
      if (!more) prefetch(0, pre[0]);            // last k-tile: the first sub-tile's epilogue inputs ride under its MFMAs
      if (!CIPS_TUNE(g.dbg & 1)) compute(kt & 1, more && !CIPS_TUNE(g.dbg & 2), (kt + 1) & 1, (kt + 1) * BK);
      else if (more && !CIPS_TUNE(g.dbg & 2)) {
        const int kb1 = prep_b(src, (kt + 1) * BK);
#pragma unroll
        for (int pc = 0; pc < 8; ++pc) dma_piece(src, pc, (kt + 1) * BK, kb1, smem + ((kt + 1) & 1) * STAGE);
      }
    }
    if (nk == 0) prefetch(0, pre[0]);
    __syncthreads();   // main-loop LDS reads are done everywhere; the scratch regions alias stage 1

    // ---------------- the next output tile's first k-tile streams into stage 0 while this tile's epilogue runs
    first_issued = false;
    if (nk > 0 && (nk & 1) == 0 && tseq + (int)gridDim.x < g.total && !CIPS_TUNE(g.dbg & 2)) {
      int tm2, tn2, bz2, kc2;
      decode(tseq + gridDim.x, tm2, tn2, bz2, kc2);
      Src nsrc;
      make_src(tm2, tn2, bz2, kc2, lane, nsrc);
      const int kbn0 = prep_b(nsrc, 0);
#pragma unroll
      for (int pc = 0; pc < 8; ++pc) dma_piece(nsrc, pc, 0, kbn0, smem);
      first_issued = true;
    }
    prefetch(1, pre[1]);
    // every global access below is younger than those DMA pieces; with the rank-3 term (scalar gathers, unknown count)
    // fall back to a full drain
    {
      // a LOWER bound is required (waiting on vmcnt(n) with n above the real count would not wait for the DMA):
      // boundary tiles may skip predicated-off accesses, so only interior tiles use the counted wait
      const int stores_per_sub = (d.C_unmasked ? 4 : 0) + (d.mask_out ? 2 : 0) + (d.C ? 4 : 0) + (Phi ? 4 : 0);
      const bool interior = (m0 + BM <= M) && (n0 + BN <= N);
      ops_after = (d.rgb_g || !interior) ? 0 : 8 * stores_per_sub + 7 * LOADS_PER_SUB;
    }

#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                         __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
    auto pack8 = [](const float (&x)[8]) -> uint4 {          // 8 floats -> 8 bf16 (RNE), v_cvt_pk_bf16_f32
      unsigned w[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        f32x2 t = {x[2 * e], x[2 * e + 1]};
        w[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
      }
      return make_uint4(w[0], w[1], w[2], w[3]);
    };
    auto unpack8 = [](const uint4& u, float (&x)[8]) {
      const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) { x[2 * e] = __uint_as_float(w[e] << 16); x[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); }
    };
    // conv bias of this lane's four output rows: requested here, first used after the first sub-tile's scratch round trip
    // (loaded at the point of use, every sub-tile waited for an L2 round trip: +3.5 ms per GAN step)
    float cbias[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    if constexpr (CONV) {
      if (g.cv.bias) {
#pragma unroll
        for (int si = 0; si < 2; ++si)
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const int row = m0 + wm * 64 + si * 32 + h_rr + 16 * c;
            cbias[si][c] = g.cv.bias[row < M ? row : M - 1];
          }
      }
    }
    static_for(std::make_integer_sequence<int, 8>{}, [&](auto ST) {
      constexpr int st = decltype(ST)::value, si = st >> 2, jj = st & 3;
      const int row0 = m0 + wm * 64 + si * 32, col0 = n0 + wn * 128 + jj * 32;
      // raw accumulators -> scratch (lane = column), read back as lane = 8 consecutive columns of rows h_rr, h_rr + 16
#pragma unroll
      for (int r = 0; r < 16; ++r) sc_f[mfma_row(r, hf) * PF + l31] = acc[si][jj][r];
      WAVE_SYNC();
      float x[2][8];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const float4 a = *reinterpret_cast<const float4*>(sc_f + (h_rr + 16 * c) * PF + h_c8);
        const float4 b = *reinterpret_cast<const float4*>(sc_f + (h_rr + 16 * c) * PF + h_c8 + 4);
        x[c][0] = a.x; x[c][1] = a.y; x[c][2] = a.z; x[c][3] = a.w; x[c][4] = b.x; x[c][5] = b.y; x[c][6] = b.z; x[c][7] = b.w;
      }
      WAVE_SYNC();                                           // scratch may be overwritten by the next sub-tile
      Pre& cur = pre[st & 1];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int row = row0 + h_rr + 16 * c, col = col0 + h_c8;
        const bool ok = row < M && col < N;
        float (&y)[8] = x[c];
        if constexpr (HAS_ADD) {
          y[0] += cur.add[2 * c].x; y[1] += cur.add[2 * c].y; y[2] += cur.add[2 * c].z; y[3] += cur.add[2 * c].w;
          y[4] += cur.add[2 * c + 1].x; y[5] += cur.add[2 * c + 1].y; y[6] += cur.add[2 * c + 1].z; y[7] += cur.add[2 * c + 1].w;
        }
        if (d.rgb_g && ok) {                                 // rank-3 term: + g[row][0..2] . rgb_w[0..2][col..col+8]
          const float* gp = d.rgb_g + ((long long)bz * M + row) * 3;
          const float g0 = gp[0], g1 = gp[1], g2 = gp[2];
#pragma unroll
          for (int e = 0; e < 8; ++e)
            y[e] = fmaf(g0, d.rgb_w[col + e], fmaf(g1, d.rgb_w[N + col + e], fmaf(g2, d.rgb_w[2 * N + col + e], y[e])));
        }
        if (d.C_unmasked && ok) {
          float* q = d.C_unmasked + cb + (long long)row * d.ldc + col;
          *reinterpret_cast<float4*>(q) = make_float4(y[0], y[1], y[2], y[3]);
          *reinterpret_cast<float4*>(q + 4) = make_float4(y[4], y[5], y[6], y[7]);
        }
        if constexpr (HAS_MASK) {
          const unsigned w[4] = {cur.mask[c].x, cur.mask[c].y, cur.mask[c].z, cur.mask[c].w};
          if (d.gate_bits & 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] *= ((w[0] >> e) & 1u) ? 1.f : d.slope;
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const unsigned mb = (e & 1) ? (w[e >> 1] >> 16) : (w[e >> 1] & 0xffffu);
              const bool pos = ((mb & 0x8000u) == 0) && ((mb & 0x7fffu) != 0);
              y[e] *= pos ? 1.f : d.slope;
            }
          }
        }
        if (d.act) {
#pragma unroll
          for (int e = 0; e < 8; ++e) y[e] = lrelu(y[e], d.slope);
        }
        const long long po = pb + (long long)row * d.ldp + col;
        if (d.mask_out && ok) {
          if (d.gate_bits & 2) {
            unsigned bits = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) bits |= (y[e] > 0.f ? 1u : 0u) << e;
            ((unsigned char*)d.mask_out)[po >> 3] = (unsigned char)bits;
          } else {
            *reinterpret_cast<uint4*>((u16*)d.mask_out + po) = pack8(y);
          }
        }
        if constexpr (HAS_RES) {
          float rh[8], rl[8];
          unpack8(cur.rh[c], rh); unpack8(cur.rl[c], rl);
#pragma unroll
          for (int e = 0; e < 8; ++e) y[e] += rh[e] + rl[e];
        }
        if constexpr (CONV) {                                // EqualConv2d + FusedLeakyReLU in one pass (unsplit launches only)
          if (g.cv.bias) {
            const float bv = cbias[si][c];
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] += bv;
          }
          if (g.cv.act) {
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = lrelu(y[e], g.cv.slope) * g.cv.act_scale;
          }
        }
        if (d.C && ok) {
          float* q = d.C + cb + (long long)row * d.ldc + col;
          *reinterpret_cast<float4*>(q) = make_float4(y[0], y[1], y[2], y[3]);
          *reinterpret_cast<float4*>(q + 4) = make_float4(y[4], y[5], y[6], y[7]);
        }
        if (Phi && ok) {
          const uint4 hi = pack8(y);
          float hf_[8], lo[8];
          unpack8(hi, hf_);
#pragma unroll
          for (int e = 0; e < 8; ++e) lo[e] = y[e] - hf_[e];
          *reinterpret_cast<uint4*>(Phi + po) = hi;
          *reinterpret_cast<uint4*>(Plo + po) = pack8(lo);
        }
      }
      if (st + 2 < 8) prefetch(st + 2, cur);                 // this sub-tile's inputs are consumed: request those of st + 2
    });
    __syncthreads();   // scratch (stage 1) is free again before the next tile's second k-tile lands in it
#undef WAVE_SYNC
  }  // persistent tile loop
}

// ------------------------------------------------------------------------------------------------------------------
// Round 3: the implicit-GEMM convolution under the two-register-set schedule of gemm_bf16x3_v3.hip (the 12 fragments of
// k-step j+1 are read between the MFMAs of k-step j; one barrier per k-tile, four MFMAs into its second k-step; the DMA of
// k-tile t+2 follows that barrier).  Same tile, LDS image, gather addressing (make_src / prep_b above) and MFMA order per
// accumulator as the CONV instantiation of the kernel above: bit-identical outputs.  A convolution's contraction is long
// (9 C / 32 k-tiles), so tiles do not overlap here: k-tiles 0 and 1 are requested at the tile start.
__device__ __forceinline__ constexpr bool cq_is_a(int q) { return q == 0 || q == 5 || q == 6 || q == 11; }
__device__ __forceinline__ constexpr int cq_off(int q) {
  return q == 0 ? OFF_ALO : q == 5 ? OFF_ALO + 32 * ROWB : q == 6 ? OFF_AHI : q == 11 ? OFF_AHI + 32 * ROWB
       : q <= 4 ? OFF_BHI + (q - 1) * 32 * ROWB : OFF_BLO + (q - 7) * 32 * ROWB;
}
__device__ __forceinline__ constexpr int cm_a(int m) { return (m >> 3) == 0 ? (((m >> 2) & 1) ? 5 : 0) : (((m >> 2) & 1) ? 11 : 6); }
__device__ __forceinline__ constexpr int cm_b(int m) { return (m >> 3) == 1 ? 7 + (m & 3) : 1 + (m & 3); }

__global__ __launch_bounds__(512) void conv2d_x3_v3_kernel(WArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const cips_gemm_x3_desc& d = g.d;
  const int tid = threadIdx.x;
  const int lane0 = tid & 63;
  const int uw = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = uw, wm = wave >> 1, wn = wave & 1;
  const int M = d.M;
  const int ksplit = g.ksplit > 1 ? g.ksplit : 1;
  const unsigned sbase = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)smem);

  for (int tseq = blockIdx.x; tseq < g.total; tseq += gridDim.x) {
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int l31 = lane & 31, hf = lane >> 5;
    int tm, tn, bz, kc;
    // geometry of this tile's range (uniform values: scalar registers)
    int N = d.N, K = d.K, lda = d.lda, ldc = d.ldc, p_kw = g.cv.kw, pad_y = g.cv.pad, pad_x = g.cv.pad, p_Wo = g.cv.Wo;
    int nimg = g.cv.nimg;
    long long a_off = 0, c_off = 0, strideC = d.strideC;
    {
      const int nx = 8;
      int q = g.total / nx, r = g.total % nx;
      int xcd = tseq % nx, idx = tseq / nx;
      int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
      int bid = base + idx;
      int tiles_n = g.tiles_n;
      if (g.nparts > 0) {
        // several ranges with different contraction lengths: every XCD takes ITS eighth of EVERY range (contiguous inside
        // the range, for L2 reuse of the operands), longest range first — one contiguous eighth of the whole sequence would
        // hand XCD 0 only the long tiles and XCD 7 only the short ones.  g.total = 8 x the longest per-XCD sequence; a slot
        // past an XCD's own count is empty.
        int left = idx, pi = 0;
        bool found = false;
        for (; pi < g.nparts; ++pi) {
          const int T = g.part[pi].ntiles, pq = T / nx, pr = T % nx;
          const int share = pq + (xcd < pr ? 1 : 0);
          if (left < share) {
            bid = ((xcd < pr) ? xcd * (pq + 1) : pr * (pq + 1) + (xcd - pr) * pq) + left;
            found = true;
            break;
          }
          left -= share;
        }
        if (!found) continue;
        const ConvPart& cp = g.part[pi];
        tiles_n = cp.tiles_n;
        N = cp.N; K = cp.K; lda = cp.lda; ldc = cp.ldc; p_kw = cp.kw; pad_y = cp.pad_y; pad_x = cp.pad_x; p_Wo = cp.Wo;
        a_off = cp.a_off; c_off = cp.c_off; strideC = cp.strideC; nimg = cp.nimg;
      }
      tn = bid % tiles_n;
      tm = (bid / tiles_n) % g.tiles_m;
      const int bk = bid / (tiles_n * g.tiles_m);
      kc = bk % ksplit;
      bz = bk / ksplit;
    }
    const int nk_all = K / BK;
    auto chunk_kt0 = [&](int kc_) -> int { return (int)((long long)kc_ * nk_all / ksplit); };
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbase = chunk_kt0(kc) * BK;
    const int nk = chunk_kt0(kc + 1) - chunk_kt0(kc);          // >= 2 (host)
    // ---- sources: weights (A) row-major [O][kh*kw*C]; pixels (B) gathered per tap from the NHWC planes of image bz
    const u16* Ahi = (const u16*)d.A_hi + a_off + (long long)m0 * lda + kbase;
    const u16* Alo = (const u16*)d.A_lo + a_off + (long long)m0 * lda + kbase;
    const u16* Bhi = (const u16*)d.B_hi + (long long)bz * g.cv.img_stride;
    const u16* Blo = (const u16*)d.B_lo + (long long)bz * g.cv.img_stride;
    const unsigned zero_rel = (unsigned)((g.cv.zero_elem - (long long)bz * g.cv.img_stride) * 2);
    unsigned offA[2], offB[2], chunk[2], imgoff[2];
    int iy0[2], ix0[2];
    {
      const int drow = lane >> 2, dslot = lane & 3;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int row = (uw + 8 * p) * 16 + drow;
        const int kcsw = dslot ^ ((row >> 2) & 3);
        const int ra = (row < M - m0) ? row : (M - m0 - 1), rb = (row < N - n0) ? row : (N - n0 - 1);
        offA[p] = (unsigned)(ra * lda + kcsw * 8) * 2u;
        int pix = n0 + rb;
        imgoff[p] = 0u;
        if (nimg > 0) {                                      // folded batch: this row's image
          const int iq = pix / nimg;
          pix -= iq * nimg;
          imgoff[p] = (unsigned)((long long)iq * g.cv.img_stride * 2);
        }
        const int oy = pix / p_Wo, ox = pix - oy * p_Wo;
        iy0[p] = oy * g.cv.stride - pad_y;
        ix0[p] = ox * g.cv.stride - pad_x;
        chunk[p] = (unsigned)kcsw * 16u;
      }
    }
    auto prep_b = [&](int k0) {                                // refresh the two per-lane B offsets for k-tile k0
      const int kk = k0 + kbase;
      const int tap = kk / g.cv.C, c0 = kk - tap * g.cv.C;
      const int ky = tap / p_kw, kx = tap - ky * p_kw;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int iy = iy0[p] + ky, ix = ix0[p] + kx;
        const bool ok = (unsigned)iy < (unsigned)g.cv.H && (unsigned)ix < (unsigned)g.cv.W;
        offB[p] = (ok ? (unsigned)(((iy * g.cv.W + ix) * g.cv.C + c0) * 2) + imgoff[p] : zero_rel) + chunk[p];
      }
    };
    auto dma = [&](const u16* plane_k, unsigned off, unsigned la) {
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(off), "s"(plane_k), "s"(la) : "memory");
    };
    auto dma_piece = [&](int pc, int k0, unsigned st) {        // uses offB of the last prep_b
      const int pp = pc >> 2, which = pc & 3;
      const unsigned la = sbase + st + (unsigned)((uw + 8 * pp) * 16 * ROWB);
      if (which == 0) dma(Ahi + k0, offA[pp], la + OFF_AHI);
      else if (which == 1) dma(Alo + k0, offA[pp], la + OFF_ALO);
      else if (which == 2) dma(Bhi, offB[pp], la + OFF_BHI);
      else dma(Blo, offB[pp], la + OFF_BLO);
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int csw = (l31 >> 2) & 3;
    const unsigned fa0 = sbase + (wm * 64 + l31) * ROWB + ((hf ^ csw) << 4), fa1 = sbase + (wm * 64 + l31) * ROWB + (((2 + hf) ^ csw) << 4);
    const unsigned fb0 = sbase + (wn * 128 + l31) * ROWB + ((hf ^ csw) << 4), fb1 = sbase + (wn * 128 + l31) * ROWB + (((2 + hf) ^ csw) << 4);
    bf16x8 F0[12], F1[12];

    auto ktile = [&](auto MODE_, int kt) {                     // MODE 0 steady, 1 next-to-last, 2 last
      constexpr int MODE = decltype(MODE_)::value;
      const unsigned cur = (unsigned)(kt & 1) * STAGE, nxt = STAGE - cur;
      unsigned a1 = fa1 + cur, b1 = fb1 + cur, a0n = fa0 + nxt, b0n = fb0 + nxt;
      asm volatile("" : "+v"(a1), "+v"(b1), "+v"(a0n), "+v"(b0n));
      if constexpr (MODE == 0) prep_b((kt + 2) * BK);
      __builtin_amdgcn_sched_barrier(0);
      static_for(std::make_integer_sequence<int, 24>{}, [&](auto M_) {
        constexpr int m = decltype(M_)::value;
        acc[(m >> 2) & 1][m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F0[cm_a(m)], F0[cm_b(m)], acc[(m >> 2) & 1][m & 3], 0, 0, 0);
        if constexpr ((m & 1) == 0) {
          constexpr int q = m >> 1;
          F1[q] = LDS_B128((cq_is_a(q) ? a1 : b1) + cq_off(q));
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      static_for(std::make_integer_sequence<int, 24>{}, [&](auto M_) {
        constexpr int m = decltype(M_)::value;
        acc[(m >> 2) & 1][m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F1[cm_a(m)], F1[cm_b(m)], acc[(m >> 2) & 1][m & 3], 0, 0, 0);
        if constexpr (m == 3) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if constexpr (MODE == 0) {
          if constexpr (m >= 4 && m <= 18 && (m & 1) == 0) dma_piece((m - 4) >> 1, (kt + 2) * BK, cur);
        }
        if constexpr (MODE <= 1) {
          if constexpr (m >= 5 && m <= 15 && (m & 1) == 1) {
            constexpr int q = m - 5;
            F0[q] = LDS_B128((cq_is_a(q) ? a0n : b0n) + cq_off(q));
            F0[q + 1] = LDS_B128((cq_is_a(q + 1) ? a0n : b0n) + cq_off(q + 1));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    };

    prep_b(0);
#pragma unroll
    for (int pc = 0; pc < 8; ++pc) dma_piece(pc, 0, 0);
    prep_b(BK);
#pragma unroll
    for (int pc = 0; pc < 8; ++pc) dma_piece(pc, BK, STAGE);
    asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int q = 0; q < 12; ++q) F0[q] = LDS_B128((cq_is_a(q) ? fa0 : fb0) + cq_off(q));
    __builtin_amdgcn_sched_barrier(0);
    for (int kt = 0; kt < nk - 2; ++kt) ktile(std::integral_constant<int, 0>{}, kt);
    ktile(std::integral_constant<int, 1>{}, nk - 2);
    ktile(std::integral_constant<int, 2>{}, nk - 1);
    __syncthreads();

    // ---- epilogue: bias + FusedLeakyReLU (unsplit launches), fp32 rows of 8 through the per-wave scratch (stage 1)
    const long long cb = c_off + ((long long)kc * d.batch + bz) * strideC;
    float* sc_f = reinterpret_cast<float*>(smem + STAGE + wave * SCR_WAVE);
    const int h_rr = lane >> 2, h_c8 = (lane & 3) * 8;
    float cbias[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    if (g.cv.bias) {
#pragma unroll
      for (int si = 0; si < 2; ++si)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int row = m0 + wm * 64 + si * 32 + h_rr + 16 * c;
          cbias[si][c] = g.cv.bias[row < M ? row : M - 1];
        }
    }
    static_for(std::make_integer_sequence<int, 8>{}, [&](auto ST) {
      constexpr int st = decltype(ST)::value, si = st >> 2, jj = st & 3;
      const int row0 = m0 + wm * 64 + si * 32, col0 = n0 + wn * 128 + jj * 32;
#pragma unroll
      for (int r = 0; r < 16; ++r) sc_f[mfma_row(r, hf) * PF + l31] = acc[si][jj][r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int row = row0 + h_rr + 16 * c, col = col0 + h_c8;
        const float4 a = *reinterpret_cast<const float4*>(sc_f + (h_rr + 16 * c) * PF + h_c8);
        const float4 b = *reinterpret_cast<const float4*>(sc_f + (h_rr + 16 * c) * PF + h_c8 + 4);
        float y[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        if (g.cv.bias) {
          const float bv = cbias[si][c];
#pragma unroll
          for (int e = 0; e < 8; ++e) y[e] += bv;
        }
        if (g.cv.act) {
#pragma unroll
          for (int e = 0; e < 8; ++e) y[e] = lrelu(y[e], g.cv.slope) * g.cv.act_scale;
        }
        if (row < M && col < N) {
          float* q = d.C + cb + (long long)row * ldc + col;
          if (nimg > 0) {                                    // folded batch: column = (image, pixel); 8 | nimg keeps the 8 of a lane in one image
            const int iq = col / nimg;
            q = d.C + cb + ((long long)iq * M + row) * nimg + (col - iq * nimg);
          }
          *reinterpret_cast<float4*>(q) = make_float4(y[0], y[1], y[2], y[3]);
          *reinterpret_cast<float4*>(q + 4) = make_float4(y[4], y[5], y[6], y[7]);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    });
    __syncthreads();
  }
}

}  // namespace

// Internal entry (called by cips_gemm_bf16x3 when the shape and the epilogue qualify): same descriptor,
// 256x256 tiles.  Returns hipErrorNotSupported for combinations it has no instantiation for.
template <bool A, bool Mk, bool R, bool CV = false>
static void launch_wide(const WArgs& g, int grid, hipStream_t stream) {
  static bool attr = false;
  CIPS_PER_DEVICE(attr, false);
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm_bf16x3_wide_kernel<A, Mk, R, CV>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    attr = true;
  }
  hipLaunchKernelGGL((gemm_bf16x3_wide_kernel<A, Mk, R, CV>), dim3(grid), dim3(512), SMEM_BYTES, stream, g);
}

static int wide_grid(int total) {
  static int ncu = 0;
  CIPS_PER_DEVICE(ncu, 0);
  if (!ncu) {
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
    if (ncu <= 0) ncu = 256;
    ncu = (ncu / 8) * 8;
  }
  return total < ncu ? total : ncu;
}

namespace {
__global__ __launch_bounds__(256) void sum_chunks_kernel(const float4* __restrict__ part, float4* __restrict__ y, int nch, long long n4,
                                                         const float* __restrict__ bias, int act, float slope, float act_scale,
                                                         int row_len4, int rows) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n4; i += gridDim.x * 256LL) {
    float4 v = part[i];
    for (int c = 1; c < nch; ++c) {
      const float4 u = part[c * n4 + i];
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    if (bias) {                                              // element i: row (i / row_len4) % rows of its image
      const float bv = bias[(int)((i / row_len4) % rows)];
      v.x += bv; v.y += bv; v.z += bv; v.w += bv;
    }
    if (act) {
      v.x = lrelu(v.x, slope) * act_scale; v.y = lrelu(v.y, slope) * act_scale;
      v.z = lrelu(v.z, slope) * act_scale; v.w = lrelu(v.w, slope) * act_scale;
    }
    y[i] = v;
  }
}
}  // namespace

// Implicit-GEMM convolution (see include/cips3d_hip.h): y[b] (O, Ho*Wo) = Wp (O, kh*kw*C) . gather(x[b])^T
extern "C" int cips_conv2d_x3(const cips_conv_x3_desc* c, cips_stream_t stream) {
  if (!c || c->B <= 0 || c->C <= 0 || c->O <= 0 || c->H <= 0 || c->W <= 0 || c->kh <= 0 || c->kw <= 0 || c->stride <= 0 || c->pad < 0)
    return (int)hipErrorInvalidValue;
  if (c->C & 31) return (int)hipErrorNotSupported;
  const int Ho = (c->H + 2 * c->pad - c->kh) / c->stride + 1, Wo = (c->W + 2 * c->pad - c->kw) / c->stride + 1;
  const long long N = (long long)Ho * Wo, K = (long long)c->kh * c->kw * c->C;
  if (Ho <= 0 || Wo <= 0 || (N & 7)) return (int)hipErrorNotSupported;
  const long long img = (long long)c->H * c->W * c->C;
  if ((img * c->B + c->C) * 2 >= 0xffffffffLL || K > 0x7fffffffLL) return (int)hipErrorNotSupported;   // 32-bit lane offsets
  WArgs g = {};
  cips_gemm_x3_desc& d = g.d;
  d.A_hi = c->w_hi; d.A_lo = c->w_lo; d.B_hi = c->x_hi; d.B_lo = c->x_lo;
  d.M = c->O; d.N = (int)N; d.K = (int)K; d.lda = (int)K; d.ldb = 0; d.strideA = 0; d.strideB = 0; d.batch = c->B;
  d.C = c->y; d.ldc = (int)N; d.strideC = (long long)c->O * N; d.slope = 0.2f;
  g.cv.C = c->C; g.cv.H = c->H; g.cv.W = c->W; g.cv.kw = c->kw; g.cv.stride = c->stride; g.cv.pad = c->pad; g.cv.Wo = Wo;
  g.cv.img_stride = img; g.cv.zero_elem = img * c->B;
  // small output planes: the batch folded into the pixel dimension (the two-register-set kernel only: every chunk of the
  // contraction needs two k-tiles; a one-k-tile problem keeps the per-image tiles of the wide kernel)
  const bool fold = N < BN && c->B > 1 && (K / 32) / (c->ksplit > 1 ? c->ksplit : 1) >= 2;
  if (fold) {
    if ((long long)c->B * N > 0x7fffffffLL) return (int)hipErrorNotSupported;
    g.cv.nimg = (int)N; d.N = (int)(c->B * N); d.batch = 1; d.strideC = (long long)c->B * c->O * N;
  }
  g.tiles_m = (d.M + BM - 1) / BM;
  g.tiles_n = (d.N + BN - 1) / BN;
  const int ks = c->ksplit > 1 ? c->ksplit : 1;
  if (ks > 1 && (!c->part || ks > (int)(K / 32))) return (int)hipErrorInvalidValue;
  if (ks > 1) d.C = c->part;                    // chunk c of image b -> part[c][b]; summed into y below
  g.ksplit = ks;
  const bool act = c->act == 1;
  if (c->act != 0 && c->act != 1) return (int)hipErrorInvalidValue;
  g.cv.bias = ks > 1 ? nullptr : c->bias; g.cv.act = (ks > 1 || !act) ? 0 : 1;
  g.cv.slope = c->slope; g.cv.act_scale = c->act_scale;
  const long long total = (long long)g.tiles_m * g.tiles_n * d.batch * ks;
  if (total > 0x7fffffffLL) return (int)hipErrorInvalidValue;
  g.total = (int)total;
  g.dbg = 0;
  const bool v3 = (K / 32) / ks >= 2;           // every chunk has at least two k-tiles (else: the one-k-tile form of the wide kernel)
  if (v3) {
    static bool attr = false;
    CIPS_PER_DEVICE(attr, false);
    if (!attr) { (void)hipFuncSetAttribute((const void*)conv2d_x3_v3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES); attr = true; }
    hipLaunchKernelGGL(conv2d_x3_v3_kernel, dim3(wide_grid(g.total)), dim3(512), SMEM_BYTES, (hipStream_t)stream, g);
  } else {
    launch_wide<false, false, false, true>(g, wide_grid(g.total), (hipStream_t)stream);
  }
  if (ks > 1) {
    const long long n4 = (long long)c->B * c->O * N / 4;        // N % 8 == 0
    const long long blocks = (n4 + 255) / 256;
    hipLaunchKernelGGL(sum_chunks_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(c->part), reinterpret_cast<float4*>(c->y), ks, n4, c->bias, act ? 1 : 0,
                       c->slope, c->act_scale, (int)(N / 4), c->O);
  }
  return CIPS_CHECK_LAUNCH();
}

// Data gradient of a stride-2, unpadded convolution as four parity sub-convolutions in one launch (include/cips3d_hip.h)
extern "C" int cips_conv2d_x3_dgrad_s2(const cips_conv_dgrad_s2_desc* c, cips_stream_t stream) {
  if (!c || !c->w_hi || !c->w_lo || !c->dy_hi || !c->dy_lo || !c->dxp || c->B <= 0 || c->C <= 0 || c->O <= 0 || c->H <= 0 || c->W <= 0 ||
      c->kh <= 0 || c->kw <= 0)
    return (int)hipErrorInvalidValue;
  if (c->H < c->kh || c->W < c->kw) return (int)hipErrorInvalidValue;
  if ((c->O & 31) || (c->C & 7)) return (int)hipErrorNotSupported;
  const int Ho = (c->H - c->kh) / 2 + 1, Wo = (c->W - c->kw) / 2 + 1;
  const long long img = (long long)Ho * Wo * c->O;
  if ((img * c->B + c->O) * 2 >= 0xffffffffLL) return (int)hipErrorNotSupported;          // 32-bit lane offsets
  WArgs g = {};
  cips_gemm_x3_desc& d = g.d;
  d.A_hi = c->w_hi; d.A_lo = c->w_lo; d.B_hi = c->dy_hi; d.B_lo = c->dy_lo;
  d.M = c->C; d.batch = c->B; d.C = c->dxp; d.slope = 0.2f;
  g.cv.C = c->O; g.cv.H = Ho; g.cv.W = Wo; g.cv.stride = 1;
  g.cv.img_stride = img; g.cv.zero_elem = img * c->B;
  g.cv.bias = nullptr; g.cv.act = 0; g.cv.slope = 0.2f; g.cv.act_scale = 1.f;
  g.tiles_m = (d.M + BM - 1) / BM;
  g.ksplit = 1;
  long long tile0 = 0;
  int np = 0;
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b) {
      const int Ta = (c->kh - a + 1) / 2, Tb = (c->kw - b + 1) / 2;     // taps ky = a, a+2, ... < kh
      const int Hs = (c->H - a + 1) / 2, Ws = (c->W - b + 1) / 2;       // input rows / columns of this parity
      if (Ta <= 0 || Tb <= 0 || Hs <= 0 || Ws <= 0) continue;          // (a 1-tap kernel has no odd class: its gradient there is zero — caller's fill)
      ConvPart& cp = g.part[np++];
      cp.kw = Tb; cp.pad_y = Ta - 1; cp.pad_x = Tb - 1; cp.Wo = Ws;
      const int np_img = (Hs * Ws + 7) & ~7;
      // the batch folded into the pixel dimension whenever a plane does not fill whole 256-pixel tiles (33 x 33 -> 1 096 pixels: 5 tiles
      // per image hold 1 280, folded 32 images take 137 tiles instead of 160; 17 x 17: 37 instead of 64)
      const bool fold = c->B > 1 && (np_img % BN) != 0;
      cp.nimg = fold ? np_img : 0;
      cp.N = fold ? np_img * c->B : np_img;
      cp.K = Ta * Tb * c->O; cp.lda = cp.K;
      if (cp.K / BK < 2) return (int)hipErrorNotSupported;
      cp.ldc = np_img; cp.strideC = (long long)c->C * np_img;
      cp.a_off = c->w_off[2 * a + b]; cp.c_off = c->out_off[2 * a + b];
      if ((cp.a_off & 7) || (cp.c_off & 3)) return (int)hipErrorInvalidValue;
      cp.tiles_n = (cp.N + BN - 1) / BN;
      const long long nt = (long long)g.tiles_m * cp.tiles_n * (fold ? 1 : c->B);
      if (nt > 0x0fffffffLL) return (int)hipErrorInvalidValue;
      cp.ntiles = (int)nt;
      tile0 += nt;
    }
  if (np == 0 || tile0 > 0x0fffffffLL) return (int)hipErrorInvalidValue;
  // longest contraction first: the persistent grid then ends on the short tiles
  for (int i = 0; i < np; ++i)
    for (int j = i + 1; j < np; ++j)
      if (g.part[j].K > g.part[i].K) { ConvPart t = g.part[i]; g.part[i] = g.part[j]; g.part[j] = t; }
  int longest = 0;                                    // slots per XCD: its share of every range
  for (int x = 0; x < 8; ++x) {
    int n = 0;
    for (int i = 0; i < np; ++i) n += g.part[i].ntiles / 8 + (x < g.part[i].ntiles % 8 ? 1 : 0);
    if (n > longest) longest = n;
  }
  g.nparts = np;
  g.total = 8 * longest;
  g.tiles_n = g.part[0].tiles_n;
  static bool attr = false;
  CIPS_PER_DEVICE(attr, false);
  if (!attr) { (void)hipFuncSetAttribute((const void*)conv2d_x3_v3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES); attr = true; }
  hipLaunchKernelGGL(conv2d_x3_v3_kernel, dim3(wide_grid(g.total)), dim3(512), SMEM_BYTES, (hipStream_t)stream, g);
  return CIPS_CHECK_LAUNCH();
}

extern "C" int cips_conv2d_x3_ksplit(int B, int O, int N, int K) {
  // chunks of the contraction that fill the chip when the output has few 256 x 256 tiles (16 x 16 planes: 64 of 256 CUs)
  const long long tiles = (N < BN && B > 1) ? (long long)((O + BM - 1) / BM) * (((long long)B * N + BN - 1) / BN)      // folded batch
                                            : (long long)((O + BM - 1) / BM) * ((N + BN - 1) / BN) * B;
  const int T = K / 32;
  int best = 1;
  long long best_cost = -1;
  for (int c = 1; c <= 8; ++c) {
    if (c > 1 && T / c < 8) break;
    const long long rounds = (tiles * c + 255) / 256;
    const long long cost = rounds * ((T + c - 1) / c * 32 + 256) + (c > 1 ? 64 * c : 0);     // + the partial-sum pass
    if (best_cost < 0 || cost < best_cost) { best = c; best_cost = cost; }
  }
  return best;
}

extern "C" CIPS_INTERNAL int cips_gemm_bf16x3_wide(const cips_gemm_x3_desc* d, cips_stream_t stream) {
  if (!d || d->M <= 0 || d->N <= 0 || d->K <= 0 || d->batch <= 0) return (int)hipErrorInvalidValue;
  if ((d->K & 31) || (d->lda & 7) || (d->ldb & 7) || (d->strideA & 7) || (d->strideB & 7)) return (int)hipErrorInvalidValue;
  if (d->T_hi || (d->N & 7) || (d->ldc & 3) || (d->strideC & 3) || (d->ldp & 7) || (d->strideP & 7)) return (int)hipErrorNotSupported;
  const bool a = d->add != nullptr, m = d->mask != nullptr, r = d->res_hi != nullptr;
  if ((a && !m) || (r && (a || m))) return (int)hipErrorNotSupported;
  if (d->gate_bits && ((d->N & 31) || (d->ldp & 31) || (d->strideP & 31))) return (int)hipErrorInvalidValue;
  WArgs g = {};
  g.d = *d;
  g.ksplit = 1;
  g.tiles_m = (d->M + BM - 1) / BM;
  g.tiles_n = (d->N + BN - 1) / BN;
  long long total = (long long)g.tiles_m * g.tiles_n * d->batch;
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
  g.dbg = 0;
#ifdef CIPS_TUNING
  g.dbg = cips_tune_env("CIPS_X3_GDBG", 0);
#endif
  const int grid = g.total < ncu ? g.total : ncu;
  hipStream_t st = (hipStream_t)stream;
  if (a) launch_wide<true, true, false>(g, grid, st);
  else if (m) launch_wide<false, true, false>(g, grid, st);
  else if (r) launch_wide<false, false, true>(g, grid, st);
  else launch_wide<false, false, false>(g, grid, st);
  return CIPS_CHECK_LAUNCH();
}
