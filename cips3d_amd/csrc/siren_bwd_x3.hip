// siren_bwd_x3.hip — fused FiLM-SIREN backward for gfx950 on the bf16 matrix cores (3-pass operand split,
// fp32 accumulate): forward recompute + data gradients + ALL weight-gradient contractions in one kernel.
//
// Backward of exp/cips3d/models/generator.py:260-317 (NeRFNetwork.forward_with_frequencies_phase_shifts)
// and exp/comm/models/film_layer.py:78-107 (autograd of FiLMLayer) — see siren.hip for the forward and the
// exact-fp32 backward ("data" pass + separate GEMMs) that this kernel replaces on the default path.
//
// Why a second kernel: the fp32 data pass has to stage h1, h2, hc, da2, dac for the weight-gradient GEMMs
// through HBM — 2 KiB per sample point, 6.4 GB per step at the headline workload — and runs its five dense
// layers on v_mfma_f32_32x32x2_f32 (157 TF peak).  Here
//   * every dense product is x = hi + lo split-bf16 on v_mfma_f32_32x32x16_bf16 (al*bh + ah*bl + ah*bh,
//     ~5e-6 relative; gradients only — the forward pass stays exact fp32),
//   * the five data layers keep siren.hip's register chain: weights are the A operand (M = features out),
//     the wave's 32 points the N dimension, so the accumulator of one layer (lane = point, 16 registers =
//     16 of a tile's 32 feature rows) is, once FiLM'd and packed to bf16 pairs, the B operand of the next
//     layer: k-step (q,t) takes registers 8t..8t+7 of tile q, i.e. features 32q+16t+4hf+{0..3, 8..11}, and
//     the A fragment is read from LDS with the same permuted k,
//   * the weight gradients  dW1 = da2^T h1, dWc = dac^T h2, dWf = dfeat^T hc  contract over POINTS, so both
//     operands are needed "feature per lane, 8 points per register group".  The packed registers are
//     written to an LDS staging image [point][feature] and read back with ds_read_b64_tr_b16 — the same
//     k-major fragment read as gemm_bf16x3.hip's K-major kernel — by all four waves, each of which owns a
//     fixed set of output tiles (112 accumulator registers) for the whole chunk.  Nothing but the final
//     per-workgroup partial dW (112 KiB) goes to HBM.
//
//   * the per-feature sums over points (FiLM phase/bias gradients, layer-0 weight gradient, sigma-head
//     weight gradient) ride on the same staged operands as one extra 32x32 accumulator tile per wave against
//     an 8-column "aux" operand [1, x, y, z, 1, dsigma, 1, 1] per point, each contraction masked to its own
//     columns — no cross-lane shuffles anywhere in the loop.
//
// LDS (all 160 KiB): W1, Wc, Wf as bf16 hi/lo images (104 KiB), the per-image FiLM vectors (4 KiB), the aux
// image (4 KiB), a 48 KiB staging buffer.  One image serves both orientations: forward fragments are two ds_read_b64 per
// plane, transposed fragments (dh = W^T d) two ds_read_b64_tr_b16.  Every image (weights and staging) is
// XOR-swizzled at 8-byte granularity by a bijection of the row index chosen so that (a) 32 lanes touching
// 32 consecutive rows at one column and (b) the transpose read's 4 rows x 64 B both cover all 64 banks.
#include "common.h"
#include "../../include/cips3d_hip.h"
#include "raygen.h"
#include <type_traits>

// wave priority of the forward chain's MFMA phases (probe builds: -DCIPS_X3_PRIO)
#ifdef CIPS_X3_PRIO
#define X3_PRIO(p) __builtin_amdgcn_s_setprio(p)
#else
#define X3_PRIO(p)
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short short4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned char uchar;

constexpr int H = 128, HC = 64, CF = 32;

// ---- LDS carve (bytes) ----
constexpr int O_W1H = 0, O_W1L = 32768;                 // [128 out][128 in] bf16, 256 B rows
constexpr int O_WCH = 65536, O_WCL = 81920;             // [64 out][128 in]
constexpr int O_WFH = 98304, O_WFL = 102400;            // [32 out][64 in], 128 B rows
constexpr int O_L0 = 106496;                            // float4[128]
constexpr int O_G1 = O_L0 + 2048, O_C1 = O_G1 + 512, O_WS = O_C1 + 512;
constexpr int O_GC = O_WS + 512, O_CC = O_GC + 256;
constexpr int O_AUX = O_CC + 256;                        // [128 points][8 bf16] hi plane, then lo plane (2 KiB each)
constexpr int O_STG = O_AUX + 4096;                      // 48 KiB staging
constexpr int STG_BYTES = 49152;
constexpr int SMEM_BYTES = O_STG + STG_BYTES;            // 163840 = the whole LDS of a CU
static_assert(O_AUX == 110592 && SMEM_BYTES == 163840, "LDS carve");

// A wave's activations in "register-chain" layout (lane = point; tile q, register r <-> feature
// 32q + (r&3) + 8(r>>2) + 4hf), packed to split bf16: dword j of tile q holds registers 2j, 2j+1, so dwords
// 2g, 2g+1 are one 8-byte LDS unit (4 consecutive features) and dwords 4t..4t+3 are the MFMA B operand of
// k-step (q,t).  Plain dword arrays on purpose: arrays of uint2 pairs defeat SROA and end up in scratch.
template <int Q> struct Act { unsigned hi[Q][8], lo[Q][8]; };

__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
  f32x2 v = {a, b};
  bf16x2 h = __builtin_convertvector(v, bf16x2);
  hi = __builtin_bit_cast(unsigned, h);
  f32x2 hf = {__uint_as_float(hi << 16), __uint_as_float(hi & 0xffff0000u)};
  f32x2 r = v - hf;
  bf16x2 l = __builtin_convertvector(r, bf16x2);
  lo = __builtin_bit_cast(unsigned, l);
}
// The same split on fp16 planes (x = hi + lo, 11 + 11 mantissa bits: 2^-22 relative where bf16 planes give 2^-17), for
// operands of known range only — the forward chain's activations are sines and its weights are staged with a per-matrix
// power-of-two scale (stage_weights_x3<PRE, true>) — fp16 has 5 exponent bits.  Same instruction count as split2.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split2h(float a, float b, unsigned& hi, unsigned& lo) {
  f32x2 v = {a, b};
  f16x2 h = __builtin_convertvector(v, f16x2);
  hi = __builtin_bit_cast(unsigned, h);
  // residual x - hi in ONE instruction per element: v_fma_mix_f32 reads the fp16 half in place (op_sel picks the half,
  // op_sel_hi marks source 0 as fp16) — hipcc's own code is v_cvt_f32_f16 x2 + v_pk_add_f32 (5 instead of 4 per pair, and
  // a packed-f32 op between MFMAs costs more than its slot, MI355X_MICROARCH.md); it has no builtin and folds
  // fma(-1, fpext(h), x) back into the subtraction.  Plain VALU -> VALU dependencies: no wait states to pad.
  float r0, r1;
  asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi), "v"(a));
  asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi), "v"(b));
  f32x2 r = {r0, r1};
  f16x2 l = __builtin_convertvector(r, f16x2);
  lo = __builtin_bit_cast(unsigned, l);
}
template <bool F16>
__device__ __forceinline__ void split2t(float a, float b, unsigned& hi, unsigned& lo) {
  if constexpr (F16) split2h(a, b, hi, lo); else split2(a, b, hi, lo);
}
// 32 values of a register group (tiles q0, q0+1) -> packed
template <int Q>
__device__ __forceinline__ void pack32(const float (&v)[32], Act<Q>& o, int q0) {
#pragma unroll
  for (int qq = 0; qq < 2; ++qq)
#pragma unroll
    for (int j = 0; j < 8; ++j) split2(v[16 * qq + 2 * j], v[16 * qq + 2 * j + 1], o.hi[q0 + qq][j], o.lo[q0 + qq][j]);
}
// Pin packed values where they are computed: hipcc otherwise sinks the whole producing computation into the
// `if (wave == turn)` staging blocks, serialising it across the workgroup's waves.
template <int Q>
__device__ __forceinline__ void pin(Act<Q>& o) {
#pragma unroll
  for (int q = 0; q < Q; ++q)
#pragma unroll
    for (int j = 0; j < 8; ++j) { asm volatile("" : "+v"(o.hi[q][j])); asm volatile("" : "+v"(o.lo[q][j])); }
}
__device__ __forceinline__ bf16x8 mk8(unsigned a, unsigned b, unsigned c, unsigned d) {
  u32x4 v = {a, b, c, d};
  return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ f32x16 x3(f32x16 acc, bf16x8 ah, bf16x8 al, bf16x8 bh, bf16x8 bl) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
  return acc;
}
// same pass order on fp16 planes (v_mfma_f32_32x32x16_f16: the bf16 instruction's rate and fragment layout)
__device__ __forceinline__ f32x16 x3h(f32x16 acc, u32x4 ah, u32x4 al, u32x4 bh, u32x4 bl) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, al), __builtin_bit_cast(f16x8, bh), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, bl), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, bh), acc, 0, 0, 0);
  return acc;
}

// LDS image layout (weights and staging alike): [column/32][row][32 columns] — 64-byte rows of 8 units
// (unit = 4 bf16 = 8 B), column blocks R*64 bytes apart, the unit index XORed with 3 bits of the row:
//   weights  (SH = 2): unit ^ ((row >> 2) & 7)     staging (SH = 1): unit ^ ((row >> 1) & 7)
// Probed on hardware (scripts/probe/lds_layout_probe.hip, SQ_LDS_BANK_CONFLICT = 0 for all three patterns):
//  * ds_read_b64_tr_b16 — a 32-lane group covers 4 rows x 64 B = one 256-B bank row whatever the in-row order;
//  * forward fragments, ds_read_b64 — 32 consecutive rows at one unit: (row & 3) picks the 64-B quarter,
//    (row >> 2) & 7 the unit inside it;
//  * staging stores, ds_write_b64 (16-lane groups, 128-B bank row) — (row & 1) picks the half, (row >> 1) & 7
//    the unit.
// and every fragment address is  lane base + compile-time immediate  (LaneAddr below).
template <int SH> __device__ __forceinline__ int img_addr(int row, int unit, int R) {
  return (unit >> 3) * R * 64 + row * 64 + (((unit & 7) ^ ((row >> SH) & 7)) << 3);
}

// All LDS traffic goes through 32-bit LDS byte addresses (lane base + compile-time constant), so that the
// constant lands in the instruction's 16-bit offset field; arithmetic on generic pointers does not fold.
#define LDS_PTR(T, a) ((__attribute__((address_space(3))) T*)(uintptr_t)(a))
__device__ __forceinline__ uint2 lds_tr(unsigned a) {
  short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(short4v, a));
  return __builtin_bit_cast(uint2, v);
}
// plain 8-byte LDS read that the load/store optimizer must not fuse into ds_read2st64_b64 (half the
// bandwidth and 2-way conflicts on this layout)
__device__ __forceinline__ uint2 lds_b64(unsigned a) {
  const unsigned long long v = *LDS_PTR(const volatile unsigned long long, a);
  return make_uint2((unsigned)v, (unsigned)(v >> 32));
}
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void lds_st64(unsigned a, unsigned x, unsigned y) { u32x2 v = {x, y}; *LDS_PTR(u32x2, a) = v; }
__device__ __forceinline__ float4 lds_ld4(unsigned a) { const f32x4 v = *LDS_PTR(const f32x4, a); return make_float4(v[0], v[1], v[2], v[3]); }

__device__ __forceinline__ constexpr int featidx(int q, int r, int hf) { return q * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf; }

template <int NM>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[NM]) {
#pragma unroll
  for (int m = 0; m < NM; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
}

struct Frag { unsigned h[4], l[4]; };   // one A (or B) fragment: 8 bf16 per plane
__device__ __forceinline__ void put(unsigned (&d)[4], int i, uint2 v) { d[i] = v.x; d[i + 1] = v.y; }

// Dense layers run as a flat list of (k-step, m-tile) items, three MFMAs each, with the A fragment of item
// i+2 requested from LDS before the MFMAs of item i issue (ring of 3 fragments = 24 registers);
// sched_barrier(0) pins that order — left alone, hipcc hoists hundreds of LDS reads and spills.
template <int NM, int KS, bool F16 = false, typename LoadF>
__device__ __forceinline__ void run_layer(LoadF load, const Act<(KS + 1) / 2>& in, f32x16 (&acc)[NM]) {
  constexpr int NI = NM * KS, D = 2;
  Frag ring[D + 1];
#pragma unroll
  for (int i = 0; i < D; ++i)
    if (i < NI) load(i / NM, i % NM, ring[i]);
#pragma unroll
  for (int it = 0; it < NI; ++it) {
    if (it + D < NI) load((it + D) / NM, (it + D) % NM, ring[(it + D) % (D + 1)]);
    const int s = it / NM, m = it % NM, q = s >> 1, t = s & 1;
    const Frag& f = ring[it % (D + 1)];
    if constexpr (F16) {
      const u32x4 bh = {in.hi[q][4 * t], in.hi[q][4 * t + 1], in.hi[q][4 * t + 2], in.hi[q][4 * t + 3]};
      const u32x4 bl = {in.lo[q][4 * t], in.lo[q][4 * t + 1], in.lo[q][4 * t + 2], in.lo[q][4 * t + 3]};
      const u32x4 ah = {f.h[0], f.h[1], f.h[2], f.h[3]}, al = {f.l[0], f.l[1], f.l[2], f.l[3]};
      __builtin_amdgcn_sched_barrier(0);
      acc[m] = x3h(acc[m], ah, al, bh, bl);
      __builtin_amdgcn_sched_barrier(0);
    } else {
    const bf16x8 bh = mk8(in.hi[q][4 * t], in.hi[q][4 * t + 1], in.hi[q][4 * t + 2], in.hi[q][4 * t + 3]);
    const bf16x8 bl = mk8(in.lo[q][4 * t], in.lo[q][4 * t + 1], in.lo[q][4 * t + 2], in.lo[q][4 * t + 3]);
    __builtin_amdgcn_sched_barrier(0);
    acc[m] = x3(acc[m], mk8(f.h[0], f.h[1], f.h[2], f.h[3]), mk8(f.l[0], f.l[1], f.l[2], f.l[3]), bh, bl);
    __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// m-major dense layer with woven side work (round 4).  scripts/probe/issue_overlap_probe.hip: with ONE wave per SIMD,
// "MFMA, 4-6 VALU, MFMA, ..." costs max(matrix pipe, VALU issue) — 16 x (MFMA, 4 v_fma) = 528 cycles against 532 for the
// MFMAs alone and 340 for the VALU alone — while "16 MFMA, then 64 v_fma" costs the sum (824): a wave's own VALU does hide
// under its own MFMAs, but only when it sits BETWEEN them in program order (an MFMA waits at issue for the pipe, and
// everything behind it waits too).  So: output tile m runs all its k-steps back to back, and after EVERY MFMA one slot
// of `side(slot)` is emitted (slot = 3 * item + pass; the callers put tile m-1's epilogue — FiLM, sine / cosine, hi / lo
// split — into the slots of tile m).  sched_barrier(0) pins the order.
template <int NM, int KS, typename LoadF, typename SideF>
__device__ __forceinline__ void run_layer_mm(LoadF load, const Act<(KS + 1) / 2>& in, f32x16 (&acc)[NM], SideF side) {
  constexpr int NI = NM * KS, D = 2;
  Frag ring[D + 1];
#pragma unroll
  for (int i = 0; i < D; ++i)
    if (i < NI) load(i % KS, i / KS, ring[i]);
#pragma unroll
  for (int it = 0; it < NI; ++it) {
    if (it + D < NI) load((it + D) % KS, (it + D) / KS, ring[(it + D) % (D + 1)]);
    const int m = it / KS, s = it % KS, q = s >> 1, t = s & 1;
    const bf16x8 bh = mk8(in.hi[q][4 * t], in.hi[q][4 * t + 1], in.hi[q][4 * t + 2], in.hi[q][4 * t + 3]);
    const bf16x8 bl = mk8(in.lo[q][4 * t], in.lo[q][4 * t + 1], in.lo[q][4 * t + 2], in.lo[q][4 * t + 3]);
    const Frag& f = ring[it % (D + 1)];
    const bf16x8 ah = mk8(f.h[0], f.h[1], f.h[2], f.h[3]), al = mk8(f.l[0], f.l[1], f.l[2], f.l[3]);
    __builtin_amdgcn_sched_barrier(0);
    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[m], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    side(3 * it);
    __builtin_amdgcn_sched_barrier(0);
    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[m], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    side(3 * it + 1);
    __builtin_amdgcn_sched_barrier(0);
    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[m], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    side(3 * it + 2);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// The DS offset field is 16 bits and the carve is 160 KiB: a region base (lane base + image offset) is made
// opaque with this so that hipcc keeps it in one register and folds only the in-region constant.
__device__ __forceinline__ unsigned opaque(unsigned v) { asm volatile("" : "+v"(v)); return v; }

// Per-lane LDS address bases (bytes, including the kernel's LDS base), recomputed every round from the
// laundered lane id.
struct LaneAddr {
  unsigned fb[2][2];   // forward fragments: [k-step parity t][second half]
  unsigned tb[2][2];   // transposed fragments, register-chain k order: [k-step parity][second half]
  unsigned sb[2];      // staging fragments, natural k order: [second half]; includes O_STG
  unsigned ab;         // aux fragments; includes O_AUX
  unsigned v16, v64;   // per-feature vectors, relative to O_L0: + 16*hf (float4 of 4 features), + 64*hf (4 float4 L0 packs)
};
__device__ __forceinline__ LaneAddr lane_addr(int lane, unsigned sbase) {
  const int l31 = lane & 31, hf = lane >> 5, s16 = lane & 15, mhalf = (lane >> 4) & 1;
  const int ul = 4 * mhalf + (s16 & 3);
  LaneAddr A;
  const int e = hf ^ ((l31 >> 2) & 7);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int sec = 0; sec < 2; ++sec) {
      A.fb[t][sec] = sbase + l31 * 64 + ((e ^ (4 * t) ^ (2 * sec)) << 3);
      A.tb[t][sec] = sbase + (4 * hf + (s16 >> 2)) * 64 + (((ul ^ hf) ^ (4 * t) ^ (2 * sec)) << 3);
    }
  const int gs = 4 * hf + (s16 >> 3);
  A.sb[0] = opaque(sbase + O_STG + (8 * hf + (s16 >> 2)) * 64 + ((ul ^ gs) << 3));
  A.sb[1] = opaque(sbase + O_STG + (8 * hf + (s16 >> 2)) * 64 + ((ul ^ gs ^ 2) << 3));
  A.ab = opaque(sbase + O_AUX + (8 * hf + (s16 >> 2)) * 16 + (s16 & 1) * 8);
  A.v16 = opaque(sbase + O_L0 + 16 * hf);
  A.v64 = opaque(sbase + O_L0 + 64 * hf);
  return A;
}

// Forward-orientation dense layer: acc[m] += W[32m + i][k] * in[k][pt]; W image (R rows = out features) at
// LDS offset IMG, lo plane PLANE bytes after the hi plane.
template <int NM, int Q, int R, int IMG, int PLANE, bool F16 = false>
__device__ __forceinline__ void layer_fwd(const LaneAddr& A, const Act<Q>& in, f32x16 (&acc)[NM]) {
  const unsigned b[2][2] = {{opaque(A.fb[0][0] + IMG), opaque(A.fb[0][1] + IMG)}, {opaque(A.fb[1][0] + IMG), opaque(A.fb[1][1] + IMG)}};
  auto load = [&](int s, int m, Frag& f) {            // k-step s = 2q+t: units 8q+4t+hf and +2 of row 32m + lane
    const int c = (s >> 1) * R * 64 + m * 2048;
    put(f.h, 0, lds_b64(b[s & 1][0] + c));
    put(f.h, 2, lds_b64(b[s & 1][1] + c));
    put(f.l, 0, lds_b64(b[s & 1][0] + c + PLANE));
    put(f.l, 2, lds_b64(b[s & 1][1] + c + PLANE));
  };
  run_layer<NM, 2 * Q, F16>(load, in, acc);
}

template <int NM, int Q, int R, int IMG, int PLANE, typename SideF>
__device__ __forceinline__ void layer_fwd_mm(const LaneAddr& A, const Act<Q>& in, f32x16 (&acc)[NM], SideF side) {
  const unsigned b[2][2] = {{opaque(A.fb[0][0] + IMG), opaque(A.fb[0][1] + IMG)}, {opaque(A.fb[1][0] + IMG), opaque(A.fb[1][1] + IMG)}};
  auto load = [&](int s, int m, Frag& f) {
    const int c = (s >> 1) * R * 64 + m * 2048;
    put(f.h, 0, lds_b64(b[s & 1][0] + c));
    put(f.h, 2, lds_b64(b[s & 1][1] + c));
    put(f.l, 0, lds_b64(b[s & 1][0] + c + PLANE));
    put(f.l, 2, lds_b64(b[s & 1][1] + c + PLANE));
  };
  run_layer_mm<NM, 2 * Q>(load, in, acc, side);
}

// Transposed dense layer: acc[m] += W[k][32m + i] * in[k][pt]  (dh = W^T d), same image, transpose reads.
// KS = k-steps (16 rows each).  The B operand's k order is the register chain's: k-step ks, element e of half
// hf <-> row 16ks + 4hf + (e&3) + 8(e>>2).
template <int NM, int KS, int R, int IMG, int PLANE>
__device__ __forceinline__ void layer_tr(const LaneAddr& A, const Act<(KS + 1) / 2>& in, f32x16 (&acc)[NM]) {
  const unsigned b[2][2] = {{opaque(A.tb[0][0] + IMG), opaque(A.tb[0][1] + IMG)}, {opaque(A.tb[1][0] + IMG), opaque(A.tb[1][1] + IMG)}};
  auto load = [&](int ks, int m, Frag& f) {
    const int c = m * R * 64 + ks * 1024;
    put(f.h, 0, lds_tr(b[ks & 1][0] + c));
    put(f.h, 2, lds_tr(b[ks & 1][1] + c + 512));
    put(f.l, 0, lds_tr(b[ks & 1][0] + c + PLANE));
    put(f.l, 2, lds_tr(b[ks & 1][1] + c + 512 + PLANE));
  };
  run_layer<NM, KS>(load, in, acc);
}

template <int NM, int KS, int R, int IMG, int PLANE, typename SideF>
__device__ __forceinline__ void layer_tr_mm(const LaneAddr& A, const Act<(KS + 1) / 2>& in, f32x16 (&acc)[NM], SideF side) {
  const unsigned b[2][2] = {{opaque(A.tb[0][0] + IMG), opaque(A.tb[0][1] + IMG)}, {opaque(A.tb[1][0] + IMG), opaque(A.tb[1][1] + IMG)}};
  auto load = [&](int ks, int m, Frag& f) {
    const int c = m * R * 64 + ks * 1024;
    put(f.h, 0, lds_tr(b[ks & 1][0] + c));
    put(f.h, 2, lds_tr(b[ks & 1][1] + c + 512));
    put(f.l, 0, lds_tr(b[ks & 1][0] + c + PLANE));
    put(f.l, 2, lds_tr(b[ks & 1][1] + c + 512 + PLANE));
  };
  run_layer_mm<NM, KS>(load, in, acc, side);
}

// write a wave's packed activations (lane = point `row` of an R-row staging image, units of 4 features);
// HI / LO: offsets of the two planes inside the staging buffer
template <int Q, int R, int HI, int LO>
__device__ __forceinline__ void stage(unsigned sbase, int row, int hf, const Act<Q>& v) {
  const unsigned rb = opaque(sbase + O_STG + row * 64);
  const int g = (row >> 1) & 7;
#pragma unroll
  for (int gg = 0; gg < 4; ++gg) {
    const unsigned o = rb + (((2 * gg + hf) ^ g) << 3);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      lds_st64(o + HI + q * R * 64, v.hi[q][2 * gg], v.hi[q][2 * gg + 1]);
      lds_st64(o + LO + q * R * 64, v.lo[q][2 * gg], v.lo[q][2 * gg + 1]);
    }
  }
}

// k-major fragment of an R-row staging image (planes at offsets HI / LO of the staging buffer; sb0 / sb1 include O_STG): lane i = feature col0 + (lane&31),
// k = points 16ks + 8hf + {0..7}
template <int R, int HI, int LO>
__device__ __forceinline__ void stg_frag(unsigned sb0, unsigned sb1, int col0, int ks, Frag& f) {
  const int c = (col0 >> 5) * R * 64 + ks * 1024;
  put(f.h, 0, lds_tr(sb0 + HI + c)); put(f.h, 2, lds_tr(sb1 + HI + c + 256));
  put(f.l, 0, lds_tr(sb0 + LO + c)); put(f.l, 2, lds_tr(sb1 + LO + c + 256));
}
__device__ __forceinline__ f32x16 x3f(f32x16 acc, const Frag& a, const Frag& b) {
  return x3(acc, mk8(a.h[0], a.h[1], a.h[2], a.h[3]), mk8(a.l[0], a.l[1], a.l[2], a.l[3]),
            mk8(b.h[0], b.h[1], b.h[2], b.h[3]), mk8(b.l[0], b.l[1], b.l[2], b.l[3]));
}

// sin / cos for the backward: one multiply to revolutions, v_fract-style reduction, hardware sin/cos.
// (|arg| is tens of radians: 3e-6 rad absolute, far inside the gradient tolerance; the exact-poly variant is
// kept for trig_mode 0.)
template <bool HW>
__device__ __forceinline__ void bsincos(float x, float* s, float* c) {
  if (HW) {
    float rv = x * CIPS_INV_2PI;
    rv = rv - rintf(rv);
    *s = __builtin_amdgcn_sinf(rv);
    *c = __builtin_amdgcn_cosf(rv);
  } else {
    sincos_reduced(reduce_2pi(x), s, c);
  }
}

// B fragment of the aux image [point][8 columns] (16 B rows per plane, no swizzle): lane j supplies column
// j & 7 of points 16ks + 8hf + {0..7}; masked to the columns one contraction owns.
__device__ __forceinline__ void aux_frag(unsigned ab, int ks, unsigned mask, Frag& f) {
  const int c = ks * 256;
  put(f.h, 0, lds_tr(ab + c)); put(f.h, 2, lds_tr(ab + c + 64));
  put(f.l, 0, lds_tr(ab + c + 2048)); put(f.l, 2, lds_tr(ab + c + 2048 + 64));
#pragma unroll
  for (int i = 0; i < 4; ++i) { f.h[i] &= mask; f.l[i] &= mask; }
}
// A x B with an exact-in-bf16 B (lo plane zero, e.g. a column of ones): two passes suffice
__device__ __forceinline__ f32x16 x2f(f32x16 acc, const Frag& a, const Frag& b) {
  const bf16x8 bh = mk8(b.h[0], b.h[1], b.h[2], b.h[3]);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mk8(a.l[0], a.l[1], a.l[2], a.l[3]), bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mk8(a.h[0], a.h[1], a.h[2], a.h[3]), bh, acc, 0, 0, 0);
  return acc;
}
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

struct BwdX3Args {
  cips_siren_weights w;
  const float* points;
  const float* dfeat;
  const float* dsigma;
  float* sred;    // [B*chunks][SRED]
  float* gpart;   // [B*chunks][GPART]
  int B, P, chunk, chunks;
  unsigned long long* prof;
  int dbg;        // timing attribution, probe builds only (-DCIPS_TUNING, env CIPS_X3_DBG): bit0/1/2 skip the dWf / dWc / dW1 phases
  RayGen rg;      // points == NULL: the points are generated from the ray parameters (point index = ray * S + s)
};
constexpr int GP_G1 = 0, GP_GC = H * H, GP_GF0 = GP_GC + HC * H, GP_GF1 = GP_GF0 + CF * HC, GPART = GP_GF1 + CF * HC;
constexpr int SRED = 4 * 32 * 8 + 8;   // per wave a 32x8 tile of column sums, then 4 per-wave sums of dsigma (+ pad)

// PRE: everything that only ever feeds a sine argument is stored divided by 2 pi — W1, Wc, the layer-0 packs and the FiLM
// offsets c1 / cc — so that gain * (W h) + c comes out in REVOLUTIONS and the sine is v_fract + v_sin with no multiply
// (the FiLM gains g1 / gc and ws stay as they are: the backward multiplies by them).  A backward that runs on these images
// carries the factor through its linear chain and removes it where it writes its partial sums (siren_bwd_x4.inc).
//
// F16 (the forward kernels, round 5): the three weight images are fp16 hi / lo planes of  W * 2^k,  k per matrix such that
// max |W| * 2^k lies in [2^13, 2^14) — every element down to 2^-17 of the largest keeps both planes normal, nothing
// overflows (fp16 max 65504), and the products hi*hi, hi*lo, lo*hi are exact in the fp32 accumulator.  The scale leaves
// through the consumers: G1 and GC hold gain * 2^-k (a power of two: exact), the colour head's 2^-k sits at O_AUX + 128
// for the kernel's output fma.  Why: sigma = ws . sin(g1 (W1 h1) + c1) is the argument of two DISCONTINUOUS consumers —
// relu(sigma + noise) in fancy_integration (pigan_utils.py:246-252) and the cdf search of sample_pdf (:164-209) — so its
// rounding decides how many samples take another branch than the fp32 reference's.  bf16 planes carry W1 h1 to ~5e-6 of
// its rms, fp16 planes to ~2e-7, the level of an fp32 fmaf chain, at the same three MFMAs per k-step.
__device__ __forceinline__ float pow2_scale_for(float m, int& k) {
  const unsigned u = __float_as_uint(m);
  const int e = (int)((u >> 23) & 0xffu) - 127;
  k = 13 - e;
  if (!(m > 0.f) || e == 128) k = 0;           // all-zero, NaN or inf weights: no scaling (the result is theirs anyway)
  k = k > 100 ? 100 : (k < -100 ? -100 : k);
  return __uint_as_float((unsigned)(k + 127) << 23);
}
template <bool PRE = false, bool F16 = false>
__device__ __forceinline__ void stage_weights_x3(uchar* sm, const cips_siren_weights& w, int b) {
  const int tid = threadIdx.x, nt = blockDim.x;
  const float pre = PRE ? CIPS_INV_2PI : 1.f;
  float s1 = 1.f, sc = 1.f, sf = 1.f, i1 = 1.f, ic = 1.f, isf = 1.f;
  if constexpr (F16) {
    // per-matrix max |W|: lane-local, wave (DPP) and workgroup (LDS words at the start of the not yet written W1 image)
    float m1 = 0.f, mc = 0.f, mf = 0.f;
    for (int i = tid; i < H * 32; i += nt) {
      const float4 v = *reinterpret_cast<const float4*>(w.w1 + 4 * i);
      m1 = fmaxf(fmaxf(m1, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    for (int i = tid; i < HC * 32; i += nt) {
      const float4 v = *reinterpret_cast<const float4*>(w.wc + 4 * i);
      mc = fmaxf(fmaxf(mc, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    for (int i = tid; i < CF * 16; i += nt) {
      const float4 v = *reinterpret_cast<const float4*>(w.wf + 4 * i);
      mf = fmaxf(fmaxf(mf, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    // NaN weights: fmaxf drops them here; they reach the planes (and every output) through the split below
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      m1 = fmaxf(m1, __shfl_xor(m1, o)); mc = fmaxf(mc, __shfl_xor(mc, o)); mf = fmaxf(mf, __shfl_xor(mf, o));
    }
    float* red = reinterpret_cast<float*>(sm + O_W1H);
    const int wv = tid >> 6, nw = nt >> 6;
    if ((tid & 63) == 0) { red[3 * wv] = m1; red[3 * wv + 1] = mc; red[3 * wv + 2] = mf; }
    __syncthreads();
    m1 = 0.f; mc = 0.f; mf = 0.f;
    for (int i = 0; i < nw; ++i) { m1 = fmaxf(m1, red[3 * i]); mc = fmaxf(mc, red[3 * i + 1]); mf = fmaxf(mf, red[3 * i + 2]); }
    __syncthreads();
    int k1, kc, kf;
    s1 = pow2_scale_for(m1 * pre, k1); sc = pow2_scale_for(mc * pre, kc); sf = pow2_scale_for(mf, kf);
    i1 = __uint_as_float((unsigned)(127 - k1) << 23); ic = __uint_as_float((unsigned)(127 - kc) << 23);
    isf = __uint_as_float((unsigned)(127 - kf) << 23);
  }
  for (int i = tid; i < H * 32; i += nt) {                  // W1: 128 rows x 32 units
    const int row = i >> 5, u = i & 31;
    float4 v = *reinterpret_cast<const float4*>(w.w1 + row * H + 4 * u);
    if (PRE) { v.x *= pre; v.y *= pre; v.z *= pre; v.w *= pre; }
    if (F16) { v.x *= s1; v.y *= s1; v.z *= s1; v.w *= s1; }
    uint2 ph, pl;
    split2t<F16>(v.x, v.y, ph.x, pl.x); split2t<F16>(v.z, v.w, ph.y, pl.y);
    const int o = img_addr<2>(row, u, H);
    *reinterpret_cast<uint2*>(sm + O_W1H + o) = ph;
    *reinterpret_cast<uint2*>(sm + O_W1L + o) = pl;
  }
  for (int i = tid; i < HC * 32; i += nt) {                 // Wc: 64 rows x 32 units
    const int row = i >> 5, u = i & 31;
    float4 v = *reinterpret_cast<const float4*>(w.wc + row * H + 4 * u);
    if (PRE) { v.x *= pre; v.y *= pre; v.z *= pre; v.w *= pre; }
    if (F16) { v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc; }
    uint2 ph, pl;
    split2t<F16>(v.x, v.y, ph.x, pl.x); split2t<F16>(v.z, v.w, ph.y, pl.y);
    const int o = img_addr<2>(row, u, HC);
    *reinterpret_cast<uint2*>(sm + O_WCH + o) = ph;
    *reinterpret_cast<uint2*>(sm + O_WCL + o) = pl;
  }
  for (int i = tid; i < CF * 16; i += nt) {                 // Wf: 32 rows x 16 units
    const int row = i >> 4, u = i & 15;
    float4 v = *reinterpret_cast<const float4*>(w.wf + row * HC + 4 * u);
    if (F16) { v.x *= sf; v.y *= sf; v.z *= sf; v.w *= sf; }
    uint2 ph, pl;
    split2t<F16>(v.x, v.y, ph.x, pl.x); split2t<F16>(v.z, v.w, ph.y, pl.y);
    const int o = img_addr<2>(row, u, CF);
    *reinterpret_cast<uint2*>(sm + O_WFH + o) = ph;
    *reinterpret_cast<uint2*>(sm + O_WFL + o) = pl;
  }
  float* L0 = reinterpret_cast<float*>(sm + O_L0);
  float* G1 = reinterpret_cast<float*>(sm + O_G1); float* C1 = reinterpret_cast<float*>(sm + O_C1);
  float* WS = reinterpret_cast<float*>(sm + O_WS);
  float* GC = reinterpret_cast<float*>(sm + O_GC); float* CC = reinterpret_cast<float*>(sm + O_CC);
  for (int f = tid; f < H; f += nt) {
    const float g0 = w.g0[b * H + f], gs = g0 * w.box_scale;
    float4 pk;
    pk.x = gs * w.w0[f * 3 + 0]; pk.y = gs * w.w0[f * 3 + 1]; pk.z = gs * w.w0[f * 3 + 2];
    pk.w = fmaf(g0, w.b0[f], w.p0[b * H + f]);
    if (PRE) { pk.x *= pre; pk.y *= pre; pk.z *= pre; pk.w *= pre; }
    reinterpret_cast<float4*>(L0)[f] = pk;
    const float g1 = w.g1[b * H + f];
    G1[f] = F16 ? g1 * i1 : g1; C1[f] = fmaf(g1, w.b1[f], w.p1[b * H + f]) * pre; WS[f] = w.ws[f];
  }
  for (int f = tid; f < HC; f += nt) {
    const float gc = w.gc[b * HC + f];
    GC[f] = F16 ? gc * ic : gc; CC[f] = fmaf(gc, w.bc[f], w.pc[b * HC + f]) * pre;
  }
  if (F16 && tid == 0) *reinterpret_cast<float*>(sm + O_AUX + 128) = isf;
}

// phase timestamps for tuning (probe builds, -DCIPS_TUNING, with CIPS_X3_PROF set): workgroup (0,0), lane 0 of each wave,
// first 8 rounds, s_memtime at each phase boundary.  The production build keeps the sched_barrier: it is part of the
// hand-pinned instruction order the kernel was tuned with.
#ifdef CIPS_TUNING
#define X3_TS(i)                                                                                   \
  __builtin_amdgcn_sched_barrier(0);                                                               \
  if (a.prof && blockIdx.x == 0 && blockIdx.y == 0 && lane0 == 0 && rnd < 8)                       \
    a.prof[(rnd * 4 + wave) * 16 + (i)] = __builtin_amdgcn_s_memtime();                            \
  __builtin_amdgcn_sched_barrier(0);
#else
#define X3_TS(i) __builtin_amdgcn_sched_barrier(0);
#endif

#include "siren_bwd_x4.inc"
#define X3F_TS(i)


// ------------------------------------------------------------------------------------------------------------------
// Forward on the same split-bf16 register chain (default; CIPS_SIREN_FWD=f32 selects siren.hip's exact fp32 MFMA
// kernel): layer 0 on the VALU, W1 / Wc / Wf on v_mfma_f32_32x32x16_bf16 in three passes, sigma as a VALU
// dot with one cross-half add.  No weight-gradient accumulators, so eight waves (two per SIMD) share the LDS images.
struct FwdX3Args {
  cips_siren_weights w;
  const float* points;     // (B, P, 3) or NULL: generated from rg (point index = ray * S + s)
  float* feat;
  float* sigma;
  float* zout;             // optional (B, P): the depth of every generated point
  RayGen rg;
  int B, P, chunk;
};

template <bool HW, bool F16>
__global__ __launch_bounds__(512) void siren_fwd_x3_kernel(FwdX3Args a) {
  extern __shared__ __attribute__((aligned(1024))) uchar smem[];
  const int b = blockIdx.y;
  stage_weights_x3<HW, F16>(smem, a.w, b);
  if (threadIdx.x < CF) reinterpret_cast<float*>(smem + O_AUX)[threadIdx.x] = a.w.bf[threadIdx.x];   // bf[32] (aux image unused here)
  __syncthreads();
  const int lane0 = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned sbase = (unsigned)(uintptr_t)((__attribute__((address_space(3))) uchar*)smem);
  const float bs = a.w.bs[0];
  const float isf = F16 ? reinterpret_cast<const float*>(smem + O_AUX)[32] : 1.f;      // 2^-k of the colour head's weight image
  const int cstart = blockIdx.x * a.chunk;
  const int cend = min(cstart + a.chunk, a.P);
  for (int pbase = cstart + wave * 32; pbase < cend; pbase += 8 * 32) {
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int l31 = lane & 31, hf = lane >> 5;
    const LaneAddr LA = lane_addr(lane, sbase);
    const int p = pbase + l31;
    const bool valid = p < cend;
    const long long gp = (long long)b * a.P + (valid ? p : cend - 1);
    float px, py, pz, zpt = 0.f;
    if (a.points) { px = a.points[gp * 3 + 0]; py = a.points[gp * 3 + 1]; pz = a.points[gp * 3 + 2]; }
    else gen_point(a.rg, b, valid ? p : cend - 1, px, py, pz, zpt);

#include "siren_fwd_chain.inc"
    if (valid) {
      float* fo = a.feat + gp * CF + 4 * hf;
      const float* bfv = reinterpret_cast<const float*>(smem + O_AUX);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 v;
        v.x = fmaf(accf[0][4 * g + 0], isf, bfv[8 * g + 4 * hf + 0]);
        v.y = fmaf(accf[0][4 * g + 1], isf, bfv[8 * g + 4 * hf + 1]);
        v.z = fmaf(accf[0][4 * g + 2], isf, bfv[8 * g + 4 * hf + 2]);
        v.w = fmaf(accf[0][4 * g + 3], isf, bfv[8 * g + 4 * hf + 3]);
        *reinterpret_cast<float4*>(fo + 8 * g) = v;
      }
      if (hf == 0) {
        a.sigma[gp] = sig;
        if (a.zout) a.zout[gp] = zpt;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Fused ray-march, non-hierarchical sampling (the headline configuration: num_steps samples per ray, no resampling):
// ray set-up + FiLM-SIREN + alpha-composite (exp/comm/comm_utils.py:365-438, 584-679; exp/cips3d/models/
// generator.py:260-317; exp/pigan/pigan_utils.py:212-273) in ONE kernel that walks the samples along the ray.
// A wave owns 32 rays (lane & 31 = ray; the two lane halves hold 16 of the 32 feature channels each) and steps
// s = 0..S-1: generate the sample point, run the register-chain MLP of siren_point_x3 (weights resident in LDS), and
// fold the sample front-to-back into the ray's running transmittance / feature / depth accumulators — z is ascending by
// construction (|jitter offset| <= half a bin), so the merge of the hierarchical path is the identity here and the
// composite needs no cross-lane traffic at all.  HBM per ray: 4 B per sample of jitter (+ 4 B of noise when
// nerf_noise > 0) in, 128 B feature + 4 B depth out = 4 S + 132 B; the (B, n, S, 3) points and the
// (B, P, 32) per-sample features never exist in HBM unless the caller asks for them (feat / sigma / z outputs: the
// training forward keeps them for the backward).  Transmittance runs in double like ATen's CPU cumprod.
struct MarchArgs {
  cips_siren_weights w;
  RayGen rg;
  const float* noise;        // (B, n, S) standard normals or NULL
  float noise_std;
  int clamp_mode, flags;     // flags: bit0 last_back, bit1 white_back
  float *fea, *depth;        // (B, n, 32), (B, n)
  float *weights;            // (B, n, S) or NULL
  float *feat, *sigma, *zout;   // per-sample outputs (B, P, 32), (B, P), (B, P) or NULL
  int B, rays_per_wg;
  const unsigned char* clamp_pin;   // optional branch masks of the relu clamp (cips_march_fwd_x3's clamp_in / clamp_out):
  unsigned char* clamp_rec;         // branch per (ray, sample) supplied / recorded; both NULL in production
  int desync;                       // probe builds: shader cycles the second wave of every SIMD starts late (0 = together)
  int one_wave;                     // probe builds: waves 4-7 leave at once (one wave per SIMD; half the rays are not marched)
  unsigned long long* prof;         // probe builds: phase timestamps of workgroup (0,0), samples 8..11
};


#undef X3F_TS
#ifdef CIPS_TUNING
#define X3F_TS(i)                                                                                  \
  __builtin_amdgcn_sched_barrier(0);                                                               \
  if (a.prof && blockIdx.x == 0 && blockIdx.y == 0 && lane0 == 0 && (s >> 2) == 2)                 \
    a.prof[(((s & 3) * 8 + wave) * 8) + (i)] = __builtin_amdgcn_s_memtime();                       \
  __builtin_amdgcn_sched_barrier(0);
#else
#define X3F_TS(i)
#endif
template <bool HW, bool DBG, bool F16>
__global__ __launch_bounds__(512) void siren_march_x3_kernel(MarchArgs a) {
  extern __shared__ __attribute__((aligned(1024))) uchar smem[];
  const int b = blockIdx.y;
  stage_weights_x3<HW, F16>(smem, a.w, b);
  if (threadIdx.x < CF) reinterpret_cast<float*>(smem + O_AUX)[threadIdx.x] = a.w.bf[threadIdx.x];
  __syncthreads();
  const int lane0 = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned sbase = (unsigned)(uintptr_t)((__attribute__((address_space(3))) uchar*)smem);
  const float bs = a.w.bs[0];
  const RayGen& g = a.rg;
  const int S = g.S, n = g.n;
  const int cstart = blockIdx.x * a.rays_per_wg;
  const int cend = min(cstart + a.rays_per_wg, n);
  const float* M = g.c2w + (long long)b * 16;
  const float* bfv = reinterpret_cast<const float*>(smem + O_AUX);
  const float isf = F16 ? bfv[32] : 1.f;        // 2^-k of the colour head's weight image
  if (CIPS_TUNE(a.one_wave) && wave >= 4) return;
  if (CIPS_TUNE(a.desync) > 0 && wave >= 4) {
    const long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < (long long)a.desync) __builtin_amdgcn_s_sleep(8);
  }
  for (int rbase = cstart + wave * 32; rbase < cend; rbase += 8 * 32) {
    const int l31s = lane0 & 31, hfs = lane0 >> 5;
    const int ray_raw = rbase + l31s;
    const bool valid = ray_raw < cend;
    const int ray = valid ? ray_raw : cend - 1;
    const long long rs = ((long long)b * n + ray) * S;        // first sample of this ray in the (B, n, S) tensors
    const RayDir d = ray_dir(g, ray);
    const bool has_jit = g.jitter != nullptr;
    float bias[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bias[r] = bfv[(r & 3) + 8 * (r >> 2) + 4 * hfs];
    float F[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) F[r] = 0.f;
    float flast[16];
    double T = 1.0;
    float depth = 0.f, wsum = 0.f, wlast = 0.f, zlast = 0.f;
    // sample s: world point + depth; the depth of sample s+1 gives delta_s
    float wx, wy, wz, zs;
    ray_point(g, M, d, g.zg[0], has_jit ? g.jitter[rs] : 0.f, has_jit, wx, wy, wz, zs);
    float u_next = (has_jit && S > 1) ? g.jitter[rs + 1] : 0.f;
#pragma unroll 1
    for (int s = 0; s < S; ++s) {
      int lane = lane0;
      asm volatile("" : "+v"(lane));
      const int hf = lane >> 5;
      const LaneAddr LA = lane_addr(lane, sbase);
      // next sample's point now (its jitter was requested one step ago), the one after that requested now
      float nx = 0.f, ny = 0.f, nz = 0.f, zn = 0.f;
      if (s + 1 < S) ray_point(g, M, d, g.zg[s + 1], u_next, has_jit, nx, ny, nz, zn);
      if (has_jit && s + 2 < S) u_next = g.jitter[rs + s + 2];
      const float nse = a.noise ? a.noise[rs + s] : 0.f;

      const float px = wx, py = wy, pz = wz;
#include "siren_fwd_chain.inc"
      float f[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) f[r] = fmaf(accf[0][r], isf, bias[r]);
      // ---- composite (pigan_utils.py:239-258): alpha = 1 - exp(-delta * clamp(sigma + noise)), w = alpha * T ----
      const float delta = (s + 1 < S) ? (zn - zs) : 1e10f;
      const float sg = a.noise ? sig + nse * a.noise_std : sig;
      float dens = (a.clamp_mode == 1) ? ((sg > 20.f) ? sg : log1pf(expf(sg))) : fmaxf(sg, 0.f);
      if (DBG && a.clamp_mode == 0) {      // the debug instantiation only: the production kernel's code is untouched
        bool pass = sg > 0.f;
        if (a.clamp_pin) pass = a.clamp_pin[rs + s] != 0;
        if (a.clamp_rec && valid && hf == 0) a.clamp_rec[rs + s] = pass ? 1 : 0;
        dens = pass ? sg : 0.f;
      }
      const float alpha = 1.f - expf(-delta * dens);
      const float w = alpha * (float)T;
      T *= (double)(1.f - alpha + 1e-10f);
#pragma unroll
      for (int r = 0; r < 16; ++r) F[r] = fmaf(w, f[r], F[r]);
      depth = fmaf(w, zs, depth);
      wsum += w;
      if (s == S - 1) {
        wlast = w; zlast = zs;
#pragma unroll
        for (int r = 0; r < 16; ++r) flast[r] = f[r];
      }
      if (valid) {
        if (a.feat) {
          float* fo = a.feat + (rs + s) * CF + 4 * hf;
#pragma unroll
          for (int gq = 0; gq < 4; ++gq)
            *reinterpret_cast<float4*>(fo + 8 * gq) = make_float4(f[4 * gq], f[4 * gq + 1], f[4 * gq + 2], f[4 * gq + 3]);
        }
        if (hf == 0) {
          if (a.sigma) a.sigma[rs + s] = sig;
          if (a.zout) a.zout[rs + s] = zs;
          if (a.weights && !(s == S - 1 && (a.flags & 1))) a.weights[rs + s] = w;
        }
      }
      wx = nx; wy = ny; wz = nz; zs = zn;
      X3F_TS(7)
      __builtin_amdgcn_sched_barrier(0);
    }
    if (a.flags & 1) {           // last_back: weights[:, :, -1] += 1 - weights_sum (pigan_utils.py:261-263)
      const float extra = 1.f - wsum;
#pragma unroll
      for (int r = 0; r < 16; ++r) F[r] = fmaf(extra, flast[r], F[r]);
      depth = fmaf(extra, zlast, depth);
      if (a.weights && valid && hfs == 0) a.weights[rs + S - 1] = wlast + extra;
    }
    if (a.flags & 2) {           // white_back: rgb + 1 - weights_sum (:266-268)
      const float extra = 1.f - wsum;
#pragma unroll
      for (int r = 0; r < 16; ++r) F[r] += extra;
    }
    if (valid) {
      float* o = a.fea + ((long long)b * n + ray) * CF + 4 * hfs;
#pragma unroll
      for (int gq = 0; gq < 4; ++gq)
        *reinterpret_cast<float4*>(o + 8 * gq) = make_float4(F[4 * gq], F[4 * gq + 1], F[4 * gq + 2], F[4 * gq + 3]);
      if (hfs == 0 && a.depth) a.depth[(long long)b * n + ray] = depth;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Finalisation of the fused backward: the per-workgroup partials (sred, gpart) -> the 16 gradient tensors autograd
// expects (FiLM gains / phases per image, weights and biases summed over the batch), in one launch instead of ~40 tiny
// torch reductions / broadcasts.  Chain rule of film_layer.py:88-107 (arg = gain * (W x + b) + phase):
//   d phase = sum_p d arg;  d gain = sum_j G[f][j] W[f][j] + b[f] d phase  (G = d arg^T h_in);  dW = sum_b gain_b G_b;
//   db = sum_b gain_b d phase_b.  Layer 0's input is box_scale * point.  Fixed summation order: chunks, then images.
// One workgroup of 128 threads per output row: [0,128) rows of dW1 (+ layer 0 and the sigma head of feature f),
// [128,192) rows of dWc, [192,224) rows of dWf (+ its bias), 224: the sigma bias.
struct FinArgs {
  cips_siren_weights w;
  const float* sred; const float* gpart;
  cips_siren_grads o;
  int B, chunks;
};

__global__ __launch_bounds__(128) void siren_bwd_finalize_kernel(FinArgs a) {
  __shared__ float cols[8];
  __shared__ float red[128];
  const int task = blockIdx.x, j = threadIdx.x;
  const int B = a.B, C = a.chunks;
  auto colsum = [&](int b, int f, int col) {       // sum over chunks of column `col` of feature row f (SRED layout)
    float s = 0.f;
    const float* p = a.sred + (long long)b * C * SRED + (f >> 5) * 256 + (f & 31) * 8 + col;
    for (int c = 0; c < C; ++c) s += p[(long long)c * SRED];
    return s;
  };
  auto block_sum = [&](float v) {
    red[j] = v;
    __syncthreads();
    for (int s_ = 64; s_ > 0; s_ >>= 1) {
      if (j < s_) red[j] += red[j + s_];
      __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
  };
  if (task < 192) {
    const bool l1 = task < 128;
    const int f = l1 ? task : task - 128;
    const int goff = l1 ? GP_G1 : GP_GC;
    const float* Wrow = (l1 ? a.w.w1 : a.w.wc) + f * H;
    const float* gain = l1 ? a.w.g1 : a.w.gc;
    const int gdim = l1 ? H : HC;
    const float bias = l1 ? a.w.b1[f] : a.w.bc[f];
    const float wj = Wrow[j];
    float acc_w = 0.f, acc_b = 0.f, acc_ws = 0.f, acc_b0 = 0.f, acc_w0[3] = {0.f, 0.f, 0.f};
    for (int b = 0; b < B; ++b) {
      float v = 0.f;
      const float* gp = a.gpart + (long long)b * C * GPART + goff + f * H + j;
      for (int c = 0; c < C; ++c) v += gp[(long long)c * GPART];
      const float gn = gain[b * gdim + f];
      acc_w = fmaf(gn, v, acc_w);
      if (j < 8) cols[j] = colsum(b, f, j);
      const float dot = block_sum(v * wj);           // (its barriers also publish cols)
      if (j == 0) {
        const float dph = l1 ? cols[4] : cols[6];
        (l1 ? a.o.dp1 : a.o.dpc)[b * gdim + f] = dph;
        (l1 ? a.o.dg1 : a.o.dgc)[b * gdim + f] = fmaf(bias, dph, dot);
        acc_b = fmaf(gn, dph, acc_b);
        if (l1) {
          // layer 0 and the sigma head ride along with feature f
          const float dp0 = cols[0], g0 = a.w.g0[b * H + f];
          float s0[3], dg0 = a.w.b0[f] * dp0;
#pragma unroll
          for (int c = 0; c < 3; ++c) { s0[c] = cols[1 + c] * a.w.box_scale; dg0 = fmaf(s0[c], a.w.w0[f * 3 + c], dg0); acc_w0[c] = fmaf(g0, s0[c], acc_w0[c]); }
          a.o.dp0[b * H + f] = dp0;
          a.o.dg0[b * H + f] = dg0;
          acc_b0 = fmaf(g0, dp0, acc_b0);
          acc_ws += cols[5];
        }
      }
      __syncthreads();
    }
    (l1 ? a.o.dw1 : a.o.dwc)[f * H + j] = acc_w;
    if (j == 0) {
      (l1 ? a.o.db1 : a.o.dbc)[f] = acc_b;
      if (l1) {
        a.o.db0[f] = acc_b0; a.o.dws[f] = acc_ws;
        a.o.dw0[f * 3 + 0] = acc_w0[0]; a.o.dw0[f * 3 + 1] = acc_w0[1]; a.o.dw0[f * 3 + 2] = acc_w0[2];
      }
    }
  } else if (task < 224) {
    const int ch = task - 192;
    if (j < HC) {
      float v = 0.f;
      for (int b = 0; b < B; ++b) {
        float vb = 0.f;
        const float* gp = a.gpart + (long long)b * C * GPART + ch * HC + j;
        for (int c = 0; c < C; ++c) vb += gp[(long long)c * GPART + GP_GF0] + gp[(long long)c * GPART + GP_GF1];
        v += vb;
      }
      a.o.dwf[ch * HC + j] = v;
    }
    if (j == 0) {
      float s = 0.f;
      for (int b = 0; b < B; ++b) s += colsum(b, ch, 7) + colsum(b, 64 + ch, 7);     // waves 0 and 2 hold the two halves
      a.o.dbf[ch] = s;
    }
  } else if (j == 0) {
    float s = 0.f;
    for (int b = 0; b < B; ++b)
      for (int c = 0; c < C; ++c) {
        const float* p = a.sred + ((long long)b * C + c) * SRED + 1024;
        s += (p[0] + p[1]) + (p[2] + p[3]);
      }
    a.o.dbs[0] = s;
  }
}

}  // namespace

// 4096-point chunks when that still fills the chip (>= 3 workgroups per CU on 256 CUs), else 2048 / 1024 / 512
static int x3_chunk(int B, int P) {
  int chunk = 4096;
  while (chunk > 512 && (long long)B * ((P + chunk - 1) / chunk) < 768) chunk >>= 1;
  return chunk;
}
extern "C" int cips_siren_bwd_x3_chunks(int B, int P) {
  const int chunk = x3_chunk(B, P);
  return (P + chunk - 1) / chunk;
}
extern "C" int cips_siren_bwd_x3_gpart(void) { return GPART; }
#ifdef CIPS_TUNING
static unsigned long long* g_prof = nullptr;
extern "C" int cips_siren_bwd_x3_prof(unsigned long long* host_out) {   // tuning aid: copies the 8x4x16 timestamps
  if (!g_prof) return (int)hipErrorNotReady;
  return (int)hipMemcpy(host_out, g_prof, 8 * 4 * 16 * 8, hipMemcpyDeviceToHost);
}
#endif
extern "C" int cips_siren_bwd_x3_sred(void) { return SRED; }


static int siren_bwd_x3_launch(const cips_siren_weights* w, const float* points, const cips_ray_params* rays,
                               const float* dfeat, const float* dsigma, float* sred, float* gpart, int B, int P,
                               cips_stream_t stream);

extern "C" int cips_siren_bwd_x3(const cips_siren_weights* w, const float* points, const float* dfeat,
                                 const float* dsigma, float* sred, float* gpart, int B, int P,
                                 cips_stream_t stream) {
  if (!points) return (int)hipErrorInvalidValue;
  return siren_bwd_x3_launch(w, points, nullptr, dfeat, dsigma, sred, gpart, B, P, stream);
}

extern "C" int cips_siren_bwd_x3_rays(const cips_siren_weights* w, const cips_ray_params* rays, const float* dfeat,
                                      const float* dsigma, float* sred, float* gpart, int B, cips_stream_t stream) {
  if (!rays) return (int)hipErrorInvalidValue;
  return siren_bwd_x3_launch(w, nullptr, rays, dfeat, dsigma, sred, gpart, B, rays->W * rays->H * rays->S, stream);
}

static int siren_bwd_x3_launch(const cips_siren_weights* w, const float* points, const cips_ray_params* rays,
                               const float* dfeat, const float* dsigma, float* sred, float* gpart, int B, int P,
                               cips_stream_t stream) {
  if (!w || !dfeat || !dsigma || !sred || !gpart || B <= 0 || P <= 0) return (int)hipErrorInvalidValue;
  BwdX3Args a;
  a.w = *w; a.points = points; a.dfeat = dfeat; a.dsigma = dsigma; a.sred = sred; a.gpart = gpart;
  a.B = B; a.P = P;
  a.rg = RayGen{};
  if (!points) { const int rc = fill_raygen(a.rg, rays); if (rc) return rc; }
  a.dbg = 0; a.prof = nullptr;
#ifdef CIPS_TUNING
  a.dbg = cips_tune_env("CIPS_X3_DBG", 0);
  static unsigned long long* prof = nullptr;
  static int want_prof = -1;
  if (want_prof < 0) {
    want_prof = cips_tune_env("CIPS_X3_PROF", 0) ? 1 : 0;
    if (want_prof && hipMalloc(&prof, 8 * 4 * 16 * 8) != hipSuccess) prof = nullptr;
  }
  a.prof = prof; g_prof = prof;
#endif
  a.chunk = x3_chunk(B, P);
  a.chunks = (P + a.chunk - 1) / a.chunk;
  dim3 grid(a.chunks, B);
  static bool attr4 = false;
  CIPS_PER_DEVICE(attr4, false);
  if (!attr4) {
    hipFuncSetAttribute((const void*)siren_bwd_x4_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    hipFuncSetAttribute((const void*)siren_bwd_x4_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    attr4 = true;
  }
  if (w->trig_mode & 1)
    hipLaunchKernelGGL(siren_bwd_x4_kernel<true>, grid, dim3(256), SMEM_BYTES, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(siren_bwd_x4_kernel<false>, grid, dim3(256), SMEM_BYTES, (hipStream_t)stream, a);
  return CIPS_CHECK_LAUNCH();
}

static int siren_fwd_x3_launch(const cips_siren_weights* w, const float* points, const cips_ray_params* rays, float* feat,
                               float* sigma, float* zout, int B, int P, cips_stream_t stream);

extern "C" int cips_siren_fwd_x3(const cips_siren_weights* w, const float* points, float* feat, float* sigma, int B, int P,
                                 cips_stream_t stream) {
  if (!points) return (int)hipErrorInvalidValue;
  return siren_fwd_x3_launch(w, points, nullptr, feat, sigma, nullptr, B, P, stream);
}

extern "C" int cips_siren_fwd_x3_rays(const cips_siren_weights* w, const cips_ray_params* rays, float* feat, float* sigma,
                                      float* zout, int B, cips_stream_t stream) {
  if (!rays) return (int)hipErrorInvalidValue;
  return siren_fwd_x3_launch(w, nullptr, rays, feat, sigma, zout, B, rays->W * rays->H * rays->S, stream);
}

static int siren_fwd_x3_launch(const cips_siren_weights* w, const float* points, const cips_ray_params* rays, float* feat,
                               float* sigma, float* zout, int B, int P, cips_stream_t stream) {
  if (!w || !feat || !sigma || B <= 0 || P <= 0) return (int)hipErrorInvalidValue;
  FwdX3Args a;
  a.w = *w; a.points = points; a.feat = feat; a.sigma = sigma; a.zout = zout; a.B = B; a.P = P;
  a.rg = RayGen{};
  if (!points) { const int rc = fill_raygen(a.rg, rays); if (rc) return rc; }
  a.chunk = 4096;
  while (a.chunk > 512 && (long long)B * ((P + a.chunk - 1) / a.chunk) < 768) a.chunk >>= 1;
  dim3 grid((P + a.chunk - 1) / a.chunk, B);
  const int smem = O_STG;            // weight images + FiLM vectors + the 4 KiB slot reused for the output bias
  // trig_mode bit 0: hardware sine; bit 1 (A/B runs only): the round-1..4 bf16 operand planes instead of fp16
  const bool hw = (w->trig_mode & 1) != 0, f16 = (w->trig_mode & 2) == 0;
  auto go = [&](auto HW_, auto F16_) {
    constexpr bool HW = decltype(HW_)::value, F16 = decltype(F16_)::value;
    static bool attr_set = false;
    CIPS_PER_DEVICE(attr_set, false);
    if (!attr_set) {
      (void)hipFuncSetAttribute((const void*)siren_fwd_x3_kernel<HW, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
      attr_set = true;
    }
    hipLaunchKernelGGL((siren_fwd_x3_kernel<HW, F16>), grid, dim3(512), smem, (hipStream_t)stream, a);
  };
  using T = std::true_type; using F = std::false_type;
  if (hw) { if (f16) go(T{}, T{}); else go(T{}, F{}); }
  else { if (f16) go(F{}, T{}); else go(F{}, F{}); }
  return CIPS_CHECK_LAUNCH();
}

#ifdef CIPS_TUNING
static unsigned long long* g_mprof = nullptr;
extern "C" int cips_march_x3_prof(unsigned long long* host_out) {       // tuning aid: copies the 4x8x8 timestamps
  if (!g_mprof) return (int)hipErrorNotReady;
  return (int)hipMemcpy(host_out, g_mprof, 4 * 8 * 8 * 8, hipMemcpyDeviceToHost);
}
#endif
extern "C" int cips_march_fwd_x3(const cips_siren_weights* w, const cips_ray_params* rays, const float* noise,
                                 float noise_std, int clamp_mode, int flags, float* fea, float* depth, float* weights,
                                 float* feat, float* sigma, float* z, int B, const unsigned char* clamp_in,
                                 unsigned char* clamp_out, cips_stream_t stream) {
  if (!w || !fea || B <= 0) return (int)hipErrorInvalidValue;
  MarchArgs a;
  a.w = *w;
  const int rc = fill_raygen(a.rg, rays);
  if (rc) return rc;
  a.noise = noise; a.noise_std = noise_std; a.clamp_mode = clamp_mode; a.flags = flags;
  a.fea = fea; a.depth = depth; a.weights = weights; a.feat = feat; a.sigma = sigma; a.zout = z; a.B = B;
  a.clamp_pin = clamp_in; a.clamp_rec = clamp_out;
  a.desync = 0; a.one_wave = 0; a.prof = nullptr;
#ifdef CIPS_TUNING
  a.desync = cips_tune_env("CIPS_X3_MDESYNC", 0);
  a.one_wave = cips_tune_env("CIPS_X3_MONE", 0);
  if (cips_tune_env("CIPS_X3_MPROF", 0)) {
    if (!g_mprof && hipMalloc(&g_mprof, 4 * 8 * 8 * 8) != hipSuccess) g_mprof = nullptr;
    a.prof = g_mprof;
  }
#endif
  // a workgroup's 8 waves take 32 rays each: 256-ray chunks keep all of them busy; halve only for small images
  a.rays_per_wg = 256;
  const int n = a.rg.n;
  dim3 grid((n + a.rays_per_wg - 1) / a.rays_per_wg, B);
  const int smem = O_STG;
  const bool dbg = a.clamp_pin || a.clamp_rec;
  const bool hw = (w->trig_mode & 1) != 0, f16 = (w->trig_mode & 2) == 0;
  auto go = [&](auto HW_, auto DBG_, auto F16_) {
    constexpr bool HW = decltype(HW_)::value, DBG = decltype(DBG_)::value, F16 = decltype(F16_)::value;
    static bool attr_set = false;
    CIPS_PER_DEVICE(attr_set, false);
    if (!attr_set) {
      (void)hipFuncSetAttribute((const void*)siren_march_x3_kernel<HW, DBG, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
      attr_set = true;
    }
    hipLaunchKernelGGL((siren_march_x3_kernel<HW, DBG, F16>), grid, dim3(512), smem, (hipStream_t)stream, a);
  };
  using T = std::true_type; using F = std::false_type;
  if (hw) {
    if (dbg) { if (f16) go(T{}, T{}, T{}); else go(T{}, T{}, F{}); }
    else { if (f16) go(T{}, F{}, T{}); else go(T{}, F{}, F{}); }
  } else {
    if (dbg) { if (f16) go(F{}, T{}, T{}); else go(F{}, T{}, F{}); }
    else { if (f16) go(F{}, F{}, T{}); else go(F{}, F{}, F{}); }
  }
  return CIPS_CHECK_LAUNCH();
}

extern "C" int cips_siren_bwd_x3_finalize(const cips_siren_weights* w, const float* sred, const float* gpart, int B,
                                          int chunks, const cips_siren_grads* out, cips_stream_t stream) {
  if (!w || !sred || !gpart || !out || B <= 0 || chunks <= 0) return (int)hipErrorInvalidValue;
  const float* const* po = reinterpret_cast<const float* const*>(out);
  for (int i = 0; i < 16; ++i) if (!po[i]) return (int)hipErrorInvalidValue;
  FinArgs a;
  a.w = *w; a.sred = sred; a.gpart = gpart; a.o = *out; a.B = B; a.chunks = chunks;
  hipLaunchKernelGGL(siren_bwd_finalize_kernel, dim3(225), dim3(128), 0, (hipStream_t)stream, a);
  return CIPS_CHECK_LAUNCH();
}
