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
// LDS (156.5 KiB of 160): W1, Wc, Wf as bf16 hi/lo images (104 KiB), the per-image FiLM vectors (4.5 KiB),
// a 48 KiB staging buffer.  One image serves both orientations: forward fragments are two ds_read_b64 per
// plane, transposed fragments (dh = W^T d) two ds_read_b64_tr_b16.  Every image (weights and staging) is
// XOR-swizzled at 8-byte granularity by a bijection of the row index chosen so that (a) 32 lanes touching
// 32 consecutive rows at one column and (b) the transpose read's 4 rows x 64 B both cover all 64 banks.
#include "common.h"
#include "../../include/cips3d_hip.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short short4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned char uchar;

constexpr int H = 128, HC = 64, CF = 32;
constexpr int RED_W = 868;

// ---- LDS carve (bytes) ----
constexpr int O_W1H = 0, O_W1L = 32768;                 // [128 out][128 in] bf16, 256 B rows
constexpr int O_WCH = 65536, O_WCL = 81920;             // [64 out][128 in]
constexpr int O_WFH = 98304, O_WFL = 102400;            // [32 out][64 in], 128 B rows
constexpr int O_L0 = 106496;                            // float4[128]
constexpr int O_G1 = O_L0 + 2048, O_C1 = O_G1 + 512, O_WS = O_C1 + 512;
constexpr int O_GC = O_WS + 512, O_CC = O_GC + 256;
constexpr int O_STG = 111616;                           // 48 KiB staging, 1 KiB aligned
constexpr int STG_BYTES = 49152;
constexpr int SMEM_BYTES = O_STG + STG_BYTES;           // 160768
static_assert(O_CC + 256 <= O_STG, "LDS carve overlap");

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
// 32 values of a register group (tiles q0, q0+1) -> packed
template <int Q>
__device__ __forceinline__ void pack32(const float (&v)[32], Act<Q>& o, int q0) {
#pragma unroll
  for (int qq = 0; qq < 2; ++qq)
#pragma unroll
    for (int j = 0; j < 8; ++j) split2(v[16 * qq + 2 * j], v[16 * qq + 2 * j + 1], o.hi[q0 + qq][j], o.lo[q0 + qq][j]);
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

// Long-lived per-point values are parked in accumulation registers by hand: the data path needs ~250 arch
// VGPRs at its widest, hipcc will not move plain floats to AGPRs before it starts spilling to scratch, and a
// scratch reload on a one-wave-per-SIMD kernel is a fully exposed ~1 us stall.
#ifndef CIPS_PARK_AGPR
__device__ __forceinline__ float park(float v) { return v; }
__device__ __forceinline__ float unpark(float a) { return a; }
#else
__device__ __forceinline__ float park(float v) { float a; asm("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(v)); return a; }
__device__ __forceinline__ float unpark(float a) { float v; asm("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a)); return v; }
#endif

// 8-byte-unit XOR swizzle of an image whose rows hold 2^N units (N = 5: 128 features, 4: 64, 3: 32)
template <int N> __device__ __forceinline__ int swz(int r);
template <> __device__ __forceinline__ int swz<5>(int r) { return ((r & 3) << 3) | ((r >> 2) & 7); }
template <> __device__ __forceinline__ int swz<4>(int r) { return (((r >> 1) & 1) << 3) | ((r >> 2) & 7); }
template <> __device__ __forceinline__ int swz<3>(int r) { return (r >> 2) & 7; }
template <int N> __device__ __forceinline__ int img_off(int row, int unit) {
  return row * (8 << N) + ((unit ^ swz<N>(row)) << 3);
}

__device__ __forceinline__ uint2 lds_tr(const uchar* p) {
  short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)p);
  return __builtin_bit_cast(uint2, v);
}

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
template <int NM, int KS, typename LoadF>
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
    const bf16x8 bh = mk8(in.hi[q][4 * t], in.hi[q][4 * t + 1], in.hi[q][4 * t + 2], in.hi[q][4 * t + 3]);
    const bf16x8 bl = mk8(in.lo[q][4 * t], in.lo[q][4 * t + 1], in.lo[q][4 * t + 2], in.lo[q][4 * t + 3]);
    const Frag& f = ring[it % (D + 1)];
    __builtin_amdgcn_sched_barrier(0);
    acc[m] = x3(acc[m], mk8(f.h[0], f.h[1], f.h[2], f.h[3]), mk8(f.l[0], f.l[1], f.l[2], f.l[3]), bh, bl);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Forward-orientation dense layer: acc[m] += W[32m + i][k] * in[k][pt], W image rows = out features.
// wl = image base + (lane&31)*256 (hi plane), gx = (swz5(lane&31) ^ hf) << 3.
template <int NM, int Q, int PLANE>
__device__ __forceinline__ void layer_fwd(const uchar* wl, int gx, const Act<Q>& in, f32x16 (&acc)[NM]) {
  auto load = [&](int s, int m, Frag& f) {
    const int o0 = ((4 * s) << 3) ^ gx, o1 = o0 ^ 16;     // k-step s = 2q+t: units 8q+4t+hf and +2
    const uchar* p = wl + m * 32 * 256;
    put(f.h, 0, *reinterpret_cast<const uint2*>(p + o0));
    put(f.h, 2, *reinterpret_cast<const uint2*>(p + o1));
    put(f.l, 0, *reinterpret_cast<const uint2*>(p + PLANE + o0));
    put(f.l, 2, *reinterpret_cast<const uint2*>(p + PLANE + o1));
  };
  run_layer<NM, 2 * Q>(load, in, acc);
}

// Transposed dense layer: acc[m] += W[k][32m + i] * in[k][pt]  (dh = W^T d), same image, transpose reads.
// N = log2(units per image row), KS = k-steps (16 rows each).  The B operand's k order is the register
// chain's: k-step ks, element e of half hf <-> row 16ks + 4hf + (e&3) + 8(e>>2).
template <int N, int NM, int KS, int PLANE>
__device__ __forceinline__ void layer_tr(const uchar* img, int lane, const Act<(KS + 1) / 2>& in, f32x16 (&acc)[NM]) {
  const int hf = lane >> 5, s16 = lane & 15, mhalf = (lane >> 4) & 1;
  const int rl = 4 * hf + (s16 >> 2), ul = 4 * mhalf + (s16 & 3);
  auto load = [&](int ks, int m, Frag& f) {
    const int r0 = 16 * ks + rl, r1 = r0 + 8;
    const int o0 = img_off<N>(r0, 8 * m + ul), o1 = img_off<N>(r1, 8 * m + ul);
    put(f.h, 0, lds_tr(img + o0));
    put(f.h, 2, lds_tr(img + o1));
    put(f.l, 0, lds_tr(img + PLANE + o0));
    put(f.l, 2, lds_tr(img + PLANE + o1));
  };
  run_layer<NM, KS>(load, in, acc);
}

// write a wave's packed activations (layout: lane = point, units of 4 features) into a staging image row
template <int N, int Q>
__device__ __forceinline__ void stage(uchar* img_hi, uchar* img_lo, int row, int hf, const Act<Q>& v) {
  const int rb = row * (8 << N), g = swz<N>(row);
#pragma unroll
  for (int q = 0; q < Q; ++q)
#pragma unroll
    for (int gg = 0; gg < 4; ++gg) {
      const int o = rb + (((8 * q + 2 * gg + hf) ^ g) << 3);
      *reinterpret_cast<uint2*>(img_hi + o) = make_uint2(v.hi[q][2 * gg], v.hi[q][2 * gg + 1]);
      *reinterpret_cast<uint2*>(img_lo + o) = make_uint2(v.lo[q][2 * gg], v.lo[q][2 * gg + 1]);
    }
}

// k-major fragment of a staging image: lane i = feature col0 + (lane&31), k = points 16ks + 8hf + {0..7}
template <int N>
__device__ __forceinline__ void stg_frag(const uchar* img_hi, const uchar* img_lo, int lane, int col0, int ks, Frag& f) {
  const int hf = lane >> 5, s16 = lane & 15, mhalf = (lane >> 4) & 1;
  const int r0 = 16 * ks + 8 * hf + (s16 >> 2), r1 = r0 + 4;
  const int u = (col0 >> 2) + 4 * mhalf + (s16 & 3);
  const int o0 = img_off<N>(r0, u), o1 = img_off<N>(r1, u);
  put(f.h, 0, lds_tr(img_hi + o0)); put(f.h, 2, lds_tr(img_hi + o1));
  put(f.l, 0, lds_tr(img_lo + o0)); put(f.l, 2, lds_tr(img_lo + o1));
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

// Transpose-reduce 32 registers across the 32 lanes of each wave half (see siren.hip).
__device__ __forceinline__ float reduce32(float (&v)[32], int lane) {
#pragma unroll
  for (int off = 16, n = 32; off >= 1; off >>= 1, n >>= 1) {
    const bool up = (lane & off) != 0;
#pragma unroll
    for (int i = 0; i < n / 2; ++i) {
      float lo = v[i], hi = v[i + n / 2];
      float send = up ? lo : hi;
      float keep = up ? hi : lo;
      v[i] = keep + __shfl_xor(send, off);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  return v[0];
}

struct BwdX3Args {
  cips_siren_weights w;
  const float* points;
  const float* dfeat;
  const float* dsigma;
  float* red;     // [B*chunks*4][868]
  float* gpart;   // [B*chunks][GPART]
  int B, P, chunk, chunks;
};
constexpr int GP_G1 = 0, GP_GC = H * H, GP_GF0 = GP_GC + HC * H, GP_GF1 = GP_GF0 + CF * HC, GPART = GP_GF1 + CF * HC;

__device__ __forceinline__ void stage_weights_x3(uchar* sm, const cips_siren_weights& w, int b) {
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int i = tid; i < H * 32; i += nt) {                  // W1: 128 rows x 32 units
    const int row = i >> 5, u = i & 31;
    const float4 v = *reinterpret_cast<const float4*>(w.w1 + row * H + 4 * u);
    uint2 ph, pl;
    split2(v.x, v.y, ph.x, pl.x); split2(v.z, v.w, ph.y, pl.y);
    const int o = img_off<5>(row, u);
    *reinterpret_cast<uint2*>(sm + O_W1H + o) = ph;
    *reinterpret_cast<uint2*>(sm + O_W1L + o) = pl;
  }
  for (int i = tid; i < HC * 32; i += nt) {                 // Wc: 64 rows x 32 units
    const int row = i >> 5, u = i & 31;
    const float4 v = *reinterpret_cast<const float4*>(w.wc + row * H + 4 * u);
    uint2 ph, pl;
    split2(v.x, v.y, ph.x, pl.x); split2(v.z, v.w, ph.y, pl.y);
    const int o = img_off<5>(row, u);
    *reinterpret_cast<uint2*>(sm + O_WCH + o) = ph;
    *reinterpret_cast<uint2*>(sm + O_WCL + o) = pl;
  }
  for (int i = tid; i < CF * 16; i += nt) {                 // Wf: 32 rows x 16 units
    const int row = i >> 4, u = i & 15;
    const float4 v = *reinterpret_cast<const float4*>(w.wf + row * HC + 4 * u);
    uint2 ph, pl;
    split2(v.x, v.y, ph.x, pl.x); split2(v.z, v.w, ph.y, pl.y);
    const int o = img_off<4>(row, u);
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
    reinterpret_cast<float4*>(L0)[f] = pk;
    const float g1 = w.g1[b * H + f];
    G1[f] = g1; C1[f] = fmaf(g1, w.b1[f], w.p1[b * H + f]); WS[f] = w.ws[f];
  }
  for (int f = tid; f < HC; f += nt) {
    const float gc = w.gc[b * HC + f];
    GC[f] = gc; CC[f] = fmaf(gc, w.bc[f], w.pc[b * HC + f]);
  }
}

template <bool HW>
__global__ __launch_bounds__(256, 1) void siren_bwd_x3_kernel(BwdX3Args a) {
  extern __shared__ __attribute__((aligned(1024))) uchar smem[];
  const int b = blockIdx.y;
  stage_weights_x3(smem, a.w, b);
  __syncthreads();

  const int lane0 = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cstart = blockIdx.x * a.chunk;
  const int cend = min(cstart + a.chunk, a.P);
  const float4* L0 = reinterpret_cast<const float4*>(smem + O_L0);
  const float* G1v = reinterpret_cast<const float*>(smem + O_G1);
  const float* C1v = reinterpret_cast<const float*>(smem + O_C1);
  const float* WSv = reinterpret_cast<const float*>(smem + O_WS);
  const float* GCv = reinterpret_cast<const float*>(smem + O_GC);
  const float* CCv = reinterpret_cast<const float*>(smem + O_CC);
  uchar* stg = smem + O_STG;

  // weight-gradient accumulators, owned per wave for the whole chunk
  f32x16 aG1[4], aGc[2], aGf[1];
  zero_acc(aG1); zero_acc(aGc); zero_acc(aGf);
  float r_da1[2] = {0.f, 0.f}, r_x[3][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
  float r_da2[2] = {0.f, 0.f}, r_dac = 0.f, r_ws[2] = {0.f, 0.f}, r_df = 0.f;

  for (int pbase = cstart; pbase < cend; pbase += 128) {
    // every LDS address below is loop-invariant; laundering the lane id keeps hipcc from hoisting a few hundred
    // of them out of the loop into live registers
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int l31 = lane & 31, hf = lane >> 5;
    const int gx = (swz<5>(l31) ^ hf) << 3;
    const uchar* W1l = smem + O_W1H + l31 * 256;
    const uchar* Wcl = smem + O_WCH + l31 * 256;
    const int p = pbase + wave * 32 + l31;
    const bool valid = p < cend;
    const long long gp = (long long)b * a.P + (valid ? p : cend - 1);
    const float px = a.points[gp * 3 + 0], py = a.points[gp * 3 + 1], pz = a.points[gp * 3 + 2];
    const float dsg = valid ? a.dsigma[gp] : 0.f;
    const int prow = wave * 32 + l31;

    // ---- layer 0 (VALU); only the packed sines are kept, and only until layer 1 has consumed them ----
    f32x16 acc[4];
    zero_acc(acc);
    {
      Act<4> h1p;
#pragma unroll
      for (int gI = 0; gI < 2; ++gI) {
        float hv[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
          const float4 pk = L0[featidx(2 * gI + (i >> 4), i & 15, 0) + 4 * hf];
          float cs;
          bsincos<HW>(fmaf(pk.x, px, fmaf(pk.y, py, fmaf(pk.z, pz, pk.w))), &hv[i], &cs);
          if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        pack32(hv, h1p, 2 * gI);
      }
      // ---- recompute layer 1 ----
      layer_fwd<4, 4, O_W1L - O_W1H>(W1l, gx, h1p, acc);
    }
    float cs2[4][16];
    Act<4> h2p;
#pragma unroll
    for (int gI = 0; gI < 2; ++gI) {
      float hv[32];
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        const int q = 2 * gI + (i >> 4), r = i & 15, f = featidx(q, r, 0) + 4 * hf;
        float cs;
        bsincos<HW>(fmaf(G1v[f], acc[q][r], C1v[f]), &hv[i], &cs);
        cs2[q][r] = park(cs);
        if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
      pack32(hv, h2p, 2 * gI);
#pragma unroll
      for (int i = 0; i < 32; ++i) hv[i] *= dsg;       // sum_p dsigma * h2  (gradient of final_layer.weight)
      r_ws[gI] += reduce32(hv, lane);
    }

    // ---- recompute colour sine layer  ----
    f32x16 accc[2];
    zero_acc(accc);
    layer_fwd<2, 4, O_WCL - O_WCH>(Wcl, gx, h2p, accc);
    float csc[2][16];
    Act<2> hcp;
    {
      float hv[32];
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        const int q = i >> 4, r = i & 15, f = featidx(q, r, 0) + 4 * hf;
        bsincos<HW>(fmaf(GCv[f], accc[q][r], CCv[f]), &hv[i], &csc[q][r]);
        if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
      pack32(hv, hcp, 0);
    }

    // ---- upstream gradient of the 32 colour features ----
    Act<1> dfp;
    {
      float v[32];
      const float* dp = a.dfeat + gp * CF + 4 * hf;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 t4 = valid ? *reinterpret_cast<const float4*>(dp + 8 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
        v[4 * g + 0] = t4.x; v[4 * g + 1] = t4.y; v[4 * g + 2] = t4.z; v[4 * g + 3] = t4.w;
        split2(t4.x, t4.y, dfp.hi[0][2 * g], dfp.lo[0][2 * g]);
        split2(t4.z, t4.w, dfp.hi[0][2 * g + 1], dfp.lo[0][2 * g + 1]);
      }
      v[16] = (hf == 0) ? dsg : 0.f;
#pragma unroll
      for (int i = 17; i < 32; ++i) v[i] = 0.f;
      r_df += reduce32(v, lane);
    }

    // ---- dWf += dfeat^T hc over the workgroup's 128 points: wave -> (hc column tile w&1, point half w>>1) ----
    {
      uchar* dfH = stg, *dfL = stg + 8192, *hcH = stg + 16384, *hcL = stg + 32768;
      stage<3, 1>(dfH, dfL, prow, hf, dfp);
      stage<4, 2>(hcH, hcL, prow, hf, hcp);
      __syncthreads();
      const int jt = wave & 1, kh = wave >> 1;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        Frag fa, fb;
        stg_frag<3>(dfH, dfL, lane, 0, 4 * kh + k, fa);
        stg_frag<4>(hcH, hcL, lane, 32 * jt, 4 * kh + k, fb);
        aGf[0] = x3f(aGf[0], fa, fb);
      }
      __syncthreads();
    }

    // ---- d hc = Wf^T dfeat  (K = 32, M = 64);  dac = d hc * cos;  dpc = gc * dac ----
    zero_acc(accc);
    layer_tr<4, 2, 2, O_WFL - O_WFH>(smem + O_WFH, lane, dfp, accc);
    Act<2> dacp, dpcp;
    {
      float v[32];
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float w_[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * g + e;
            v[16 * q + r] = accc[q][r] * csc[q][r];
            w_[e] = GCv[featidx(q, r, 0) + 4 * hf] * v[16 * q + r];
          }
          split2(v[16 * q + 4 * g], v[16 * q + 4 * g + 1], dacp.hi[q][2 * g], dacp.lo[q][2 * g]);
          split2(v[16 * q + 4 * g + 2], v[16 * q + 4 * g + 3], dacp.hi[q][2 * g + 1], dacp.lo[q][2 * g + 1]);
          split2(w_[0], w_[1], dpcp.hi[q][2 * g], dpcp.lo[q][2 * g]);
          split2(w_[2], w_[3], dpcp.hi[q][2 * g + 1], dpcp.lo[q][2 * g + 1]);
          __builtin_amdgcn_sched_barrier(0);
        }
      r_dac += reduce32(v, lane);
    }

    // ---- dWc += dac^T h2: two sub-phases of 64 points; wave -> (dac row tile w&1, h2 column tiles 2(w>>1)+{0,1}) ----
    {
      uchar* daH = stg, *daL = stg + 8192, *h2H = stg + 16384, *h2L = stg + 32768;
      const int it = wave & 1, jt0 = 2 * (wave >> 1);
#pragma unroll
      for (int sp = 0; sp < 2; ++sp) {
        if ((wave >> 1) == sp) {
          stage<4, 2>(daH, daL, (wave & 1) * 32 + l31, hf, dacp);
          stage<5, 4>(h2H, h2L, (wave & 1) * 32 + l31, hf, h2p);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          Frag fa, fb0, fb1;
          stg_frag<4>(daH, daL, lane, 32 * it, k, fa);
          stg_frag<5>(h2H, h2L, lane, 32 * jt0, k, fb0);
          stg_frag<5>(h2H, h2L, lane, 32 * jt0 + 32, k, fb1);
          __builtin_amdgcn_sched_barrier(0);
          aGc[0] = x3f(aGc[0], fa, fb0);
          aGc[1] = x3f(aGc[1], fa, fb1);
          __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
      }
    }

    // ---- d h2 = Wc^T dpc + ws * dsigma  (K = 64, M = 128);  da2 = d h2 * cos;  dp2 = g1 * da2 ----
    zero_acc(acc);
    layer_tr<5, 4, 4, O_WCL - O_WCH>(smem + O_WCH, lane, dpcp, acc);
    Act<4> da2p;
    {
      Act<4> dp2p;
#pragma unroll
      for (int gI = 0; gI < 2; ++gI) {
        float v[32];
#pragma unroll
        for (int qq = 0; qq < 2; ++qq)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int q = 2 * gI + qq;
            float w_[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int r = 4 * g + e, f = featidx(q, r, 0) + 4 * hf;
              v[16 * qq + r] = fmaf(WSv[f], dsg, acc[q][r]) * unpark(cs2[q][r]);
              w_[e] = G1v[f] * v[16 * qq + r];
            }
            split2(v[16 * qq + 4 * g], v[16 * qq + 4 * g + 1], da2p.hi[q][2 * g], da2p.lo[q][2 * g]);
            split2(v[16 * qq + 4 * g + 2], v[16 * qq + 4 * g + 3], da2p.hi[q][2 * g + 1], da2p.lo[q][2 * g + 1]);
            split2(w_[0], w_[1], dp2p.hi[q][2 * g], dp2p.lo[q][2 * g]);
            split2(w_[2], w_[3], dp2p.hi[q][2 * g + 1], dp2p.lo[q][2 * g + 1]);
            __builtin_amdgcn_sched_barrier(0);
          }
        r_da2[gI] += reduce32(v, lane);
      }
      // ---- d h1 = W1^T dp2  (K = 128, M = 128) ----
      zero_acc(acc);
      layer_tr<5, 4, 8, O_W1L - O_W1H>(smem + O_W1H, lane, dp2p, acc);
    }
    // ---- da1 = d h1 * cos(layer-0 argument); the layer-0 sines are recomputed alongside for dW1 ----
    Act<4> h1p;
#pragma unroll
    for (int gI = 0; gI < 2; ++gI) {
      float v[32], t[32];
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        const float4 pk = L0[featidx(2 * gI + (i >> 4), i & 15, 0) + 4 * hf];
        float cs;
        bsincos<HW>(fmaf(pk.x, px, fmaf(pk.y, py, fmaf(pk.z, pz, pk.w))), &t[i], &cs);
        v[i] = acc[2 * gI + (i >> 4)][i & 15] * cs;
        if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
      pack32(t, h1p, 2 * gI);
#pragma unroll
      for (int i = 0; i < 32; ++i) t[i] = v[i];
      r_da1[gI] += reduce32(t, lane);
#pragma unroll
      for (int i = 0; i < 32; ++i) t[i] = v[i] * px;
      r_x[0][gI] += reduce32(t, lane);
#pragma unroll
      for (int i = 0; i < 32; ++i) t[i] = v[i] * py;
      r_x[1][gI] += reduce32(t, lane);
#pragma unroll
      for (int i = 0; i < 32; ++i) t[i] = v[i] * pz;
      r_x[2][gI] += reduce32(t, lane);
    }

    // ---- dW1 += da2^T h1: four sub-phases of 32 points; wave -> da2 row tile w, all four h1 column tiles ----
    {
      uchar* daH = stg, *daL = stg + 8192, *h1H = stg + 16384, *h1L = stg + 24576;
#pragma unroll 1
      for (int sp = 0; sp < 4; ++sp) {
        if (wave == sp) {
          stage<5, 4>(daH, daL, l31, hf, da2p);
          stage<5, 4>(h1H, h1L, l31, hf, h1p);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          Frag fa;
          stg_frag<5>(daH, daL, lane, 32 * wave, k, fa);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            Frag fb;
            stg_frag<5>(h1H, h1L, lane, 32 * j, k, fb);
            aG1[j] = x3f(aG1[j], fa, fb);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
      }
    }
  }

  // ---- write this wave's partial reductions (same row format as the fp32 data pass) ----
  const int lane = lane0, l31 = lane & 31, hf = lane >> 5;
  {
    float* row = a.red + ((long long)(b * a.chunks + blockIdx.x) * 4 + wave) * RED_W;
#pragma unroll
    for (int gI = 0; gI < 2; ++gI) {
      const int f = featidx(2 * gI + (l31 >> 4), l31 & 15, hf);
      row[f] = r_da1[gI];
      row[128 + f] = r_x[0][gI];
      row[256 + f] = r_x[1][gI];
      row[384 + f] = r_x[2][gI];
      row[512 + f] = r_da2[gI];
      row[704 + f] = r_ws[gI];
    }
    row[640 + featidx(l31 >> 4, l31 & 15, hf)] = r_dac;
    if (l31 < 16) row[832 + mfma_row(l31, hf)] = r_df;
    if (l31 == 16 && hf == 0) row[864] = r_df;
    if (l31 >= 17 && l31 < 20 && hf == 0) row[864 + (l31 - 16)] = 0.f;
  }
  // ---- write the workgroup's partial weight gradients ----
  {
    float* gp_ = a.gpart + (long long)(b * a.chunks + blockIdx.x) * GPART;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) gp_[GP_G1 + (32 * wave + mfma_row(r, hf)) * H + 32 * j + l31] = aG1[j][r];
    const int it = wave & 1, jt0 = 2 * (wave >> 1);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) gp_[GP_GC + (32 * it + mfma_row(r, hf)) * H + 32 * (jt0 + j) + l31] = aGc[j][r];
    float* gf = gp_ + ((wave >> 1) ? GP_GF1 : GP_GF0);
#pragma unroll
    for (int r = 0; r < 16; ++r) gf[mfma_row(r, hf) * HC + 32 * (wave & 1) + l31] = aGf[0][r];
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

extern "C" int cips_siren_bwd_x3(const cips_siren_weights* w, const float* points, const float* dfeat,
                                 const float* dsigma, float* red, float* gpart, int B, int P,
                                 cips_stream_t stream) {
  if (!w || !points || !dfeat || !dsigma || !red || !gpart || B <= 0 || P <= 0) return (int)hipErrorInvalidValue;
  BwdX3Args a;
  a.w = *w; a.points = points; a.dfeat = dfeat; a.dsigma = dsigma; a.red = red; a.gpart = gpart;
  a.B = B; a.P = P;
  a.chunk = x3_chunk(B, P);
  a.chunks = (P + a.chunk - 1) / a.chunk;
  dim3 grid(a.chunks, B);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)siren_bwd_x3_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    hipFuncSetAttribute((const void*)siren_bwd_x3_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    attr_set = true;
  }
  if (w->trig_mode == 1)
    hipLaunchKernelGGL(siren_bwd_x3_kernel<true>, grid, dim3(256), SMEM_BYTES, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(siren_bwd_x3_kernel<false>, grid, dim3(256), SMEM_BYTES, (hipStream_t)stream, a);
  return CIPS_CHECK_LAUNCH();
}
