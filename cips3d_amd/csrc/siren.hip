// siren.hip — fused FiLM-SIREN NeRF point MLP for gfx950 (forward, and the "data" half of
// the backward with in-kernel forward recompute).
//
// Replaces exp/cips3d/models/generator.py:260-317 (NeRFNetwork.forward_with_frequencies_
// phase_shifts), exp/comm/models/film_layer.py:78-107 (FiLMLayer.forward) and
// exp/comm/models/nerf_network.py:39-45 (UniformBoxWarp):
//   x0 = p * (2/0.24)
//   h1 = sin(g0 * (W0 x0 + b0) + p0)        3 -> 128   (VALU)
//   h2 = sin(g1 * (W1 h1 + b1) + p1)      128 -> 128   (MFMA)
//   sigma = ws . h2 + bs                  128 -> 1     (VALU dot + one cross-half add)
//   hc = sin(gc * (Wc h2 + bc) + pc)      128 -> 64    (MFMA)
//   feat = Wf hc + bf                      64 -> 32    (MFMA)
//
// MI355X mapping.  One wave owns a tile of 32 sample points.  All three dense layers run on
// v_mfma_f32_32x32x2_f32 with the WEIGHTS as the A operand (M = output features) and the
// POINTS as the B/N dimension, so the accumulator of one layer — lane = point (lane&31),
// 16 registers = 16 of the tile's 32 feature rows, the lane's upper-half flag selecting which
// 16 — is, after the FiLM sine, directly the B operand of the next layer: for k-step s the
// two lane halves supply features F(s,0) and F(s,1) from their own registers and the A
// fragment is read from LDS with the same permuted k index.  Activations never leave
// registers and never touch LDS; the fp32 weights (W1 64 KB, Wc 32 KB, Wf 8 KB) stay resident
// in LDS (108 KB of the CU's 160 KB) in odd-stride images that make every A-fragment
// ds_read_b32 conflict-free for both the forward (row = out feature) and the transposed
// backward (row = in feature) access.  fp32 MFMA issues once per 64 cycles per SIMD, so the
// single ds_read_b32 per MFMA is a few % of LDS bandwidth; the kernel is MFMA-issue bound
// (27 136 MAC per point = 832 SIMD-cycles per point at peak) with the ~320 sines per point
// on the VALU overlapping the other wave of the SIMD.
#include "common.h"
#include "../../include/cips3d_hip.h"

namespace {

constexpr int H = 128;    // hidden width
constexpr int HC = 64;    // colour hidden width
constexpr int CF = 32;    // feature channels
constexpr int LD1 = H + 1;   // W1 image stride  [128][129]
constexpr int LDC = H + 1;   // Wc image stride  [64][129]
constexpr int LDF = HC + 1;  // Wf image stride  [32][65]
constexpr int RED_W = 868;   // reduction row width (see cips3d_hip.h)

// LDS carve (floats)
constexpr int OFF_W1 = 0;
constexpr int OFF_WC = OFF_W1 + H * LD1;          // 16512
constexpr int OFF_WF = OFF_WC + HC * LDC;         // 24768
constexpr int OFF_L0 = OFF_WF + CF * LDF;         // 26848  float4[128] (16 B aligned: 26848*4 % 16 == 0)
constexpr int OFF_G1 = OFF_L0 + 4 * H;            // g1[128]
constexpr int OFF_C1 = OFF_G1 + H;                // c1[128] = g1*b1 + p1
constexpr int OFF_GC = OFF_C1 + H;                // gc[64]
constexpr int OFF_CC = OFF_GC + HC;               // cc[64]
constexpr int OFF_WS = OFF_CC + HC;               // ws[128]
constexpr int OFF_BF = OFF_WS + H;                // bf[32]
constexpr int OFF_BS = OFF_BF + CF;               // bs
constexpr int SMEM_FLOATS = OFF_BS + 4;
static_assert((OFF_L0 * 4) % 16 == 0, "L0 pack must be 16-byte aligned");

// feature index inside a layer for tile q, accumulator register r, lane half hf
__device__ __forceinline__ constexpr int featidx(int q, int r, int hf) {
  return q * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf;
}


// One dense layer on the matrix cores with explicit one-step operand prefetch.
// acc[m] += sum_{q,r} Wp[m*mstride + featidx(q,r,0)*kstride] * hin[q][r]
// sched_barrier(0) per k-step pins the schedule (4 ds_read_b32 of step s+1, then the MFMAs of
// step s); without it hipcc hoists hundreds of LDS reads and spills.
template <int NM, int Q>
__device__ __forceinline__ void mfma_layer(const float* Wp, int mstride, int kstride,
                                           const float (&hin)[Q][16], f32x16 (&acc)[NM]) {
  float an[NM];
#pragma unroll
  for (int m = 0; m < NM; ++m) an[m] = Wp[m * mstride + featidx(0, 0, 0) * kstride];
#pragma unroll
  for (int q = 0; q < Q; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float ac[NM];
#pragma unroll
      for (int m = 0; m < NM; ++m) ac[m] = an[m];
      const int s1 = q * 16 + r + 1;
      if (s1 < Q * 16) {
        const int k1 = featidx(s1 >> 4, s1 & 15, 0);
#pragma unroll
        for (int m = 0; m < NM; ++m) an[m] = Wp[m * mstride + k1 * kstride];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < NM; ++m) acc[m] = mfma32(ac[m], hin[q][r], acc[m]);
      __builtin_amdgcn_sched_barrier(0);
    }
}

template <int NM>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[NM]) {
#pragma unroll
  for (int m = 0; m < NM; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
}

__device__ __forceinline__ void stage_weights(float* sm, const cips_siren_weights& w, int b) {
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int i = tid; i < H * H; i += nt) sm[OFF_W1 + (i >> 7) * LD1 + (i & 127)] = w.w1[i];
  for (int i = tid; i < HC * H; i += nt) sm[OFF_WC + (i >> 7) * LDC + (i & 127)] = w.wc[i];
  for (int i = tid; i < CF * HC; i += nt) sm[OFF_WF + (i >> 6) * LDF + (i & 63)] = w.wf[i];
  for (int f = tid; f < H; f += nt) {
    float g0 = w.g0[b * H + f];
    float gs = g0 * w.box_scale;
    float4 pk;
    pk.x = gs * w.w0[f * 3 + 0];
    pk.y = gs * w.w0[f * 3 + 1];
    pk.z = gs * w.w0[f * 3 + 2];
    pk.w = fmaf(g0, w.b0[f], w.p0[b * H + f]);
    reinterpret_cast<float4*>(sm + OFF_L0)[f] = pk;
    float g1 = w.g1[b * H + f];
    sm[OFF_G1 + f] = g1;
    sm[OFF_C1 + f] = fmaf(g1, w.b1[f], w.p1[b * H + f]);
    sm[OFF_WS + f] = w.ws[f];
  }
  for (int f = tid; f < HC; f += nt) {
    float gc = w.gc[b * HC + f];
    sm[OFF_GC + f] = gc;
    sm[OFF_CC + f] = fmaf(gc, w.bc[f], w.pc[b * HC + f]);
  }
  for (int f = tid; f < CF; f += nt) sm[OFF_BF + f] = w.bf[f];
  if (tid == 0) sm[OFF_BS] = w.bs[0];
}

struct FwdArgs {
  cips_siren_weights w;
  const float* points;
  float* feat;
  float* sigma;
  int B, P, chunk;
};

// layer 0 on the VALU: 64 of the 128 features per lane, produced directly in layout "N"
template <bool HW, bool WITH_COS>
__device__ __forceinline__ void layer0(const float4* L0, int hf, float px, float py, float pz,
                                       float (&h)[4][16], float (&cs)[4][16]) {
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 4 * g + i;
        float4 pk = L0[featidx(q, r, 0) + 4 * hf];
        float arg = fmaf(pk.x, px, fmaf(pk.y, py, fmaf(pk.z, pz, pk.w)));
        if (WITH_COS) film_sincos<HW>(arg, &h[q][r], &cs[q][r]);
        else h[q][r] = film_sin<HW>(arg);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
}

// FiLM sine epilogue of a dense layer: h = sin(g[f] * acc + c[f])  (c = g*b + phase)
template <bool HW, bool WITH_COS, int Q>
__device__ __forceinline__ void film_act(const f32x16 (&acc)[Q], const float* gv, const float* cv,
                                         int hf, float (&h)[Q][16], float (&cs)[Q][16]) {
#pragma unroll
  for (int q = 0; q < Q; ++q)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 4 * g + i;
        const int f = featidx(q, r, 0) + 4 * hf;
        float arg = fmaf(gv[f], acc[q][r], cv[f]);
        if (WITH_COS) film_sincos<HW>(arg, &h[q][r], &cs[q][r]);
        else h[q][r] = film_sin<HW>(arg);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
}

template <bool HW>
__global__ __launch_bounds__(512, 2) void siren_fwd_kernel(FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int b = blockIdx.y;
  stage_weights(sm, a.w, b);
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int l31 = lane & 31, hf = lane >> 5;
  const int cstart = blockIdx.x * a.chunk;
  const int cend = min(cstart + a.chunk, a.P);
  const float* W1s = sm + OFF_W1 + l31 * LD1 + 4 * hf;
  const float* Wcs = sm + OFF_WC + l31 * LDC + 4 * hf;
  const float* Wfs = sm + OFF_WF + l31 * LDF + 4 * hf;
  const float4* L0 = reinterpret_cast<const float4*>(sm + OFF_L0);

  for (int p0 = cstart + wave * 32; p0 < cend; p0 += nwaves * 32) {
    const int p = p0 + l31;
    const bool valid = p < cend;
    const long long gp = (long long)b * a.P + (valid ? p : cend - 1);
    const float px = a.points[gp * 3 + 0], py = a.points[gp * 3 + 1], pz = a.points[gp * 3 + 2];

    float h[4][16];
    layer0<HW, false>(L0, hf, px, py, pz, h, h);

    // ---- layer 1 (MFMA 128x128) ----
    f32x16 acc[4];
    zero_acc(acc);
    mfma_layer<4, 4>(W1s, 32 * LD1, 1, h, acc);
    film_act<HW, false, 4>(acc, sm + OFF_G1, sm + OFF_C1, hf, h, h);

    float sig = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) sig = fmaf(sm[OFF_WS + featidx(q, r, 0) + 4 * hf], h[q][r], sig);
    sig += __shfl_xor(sig, 32);
    sig += sm[OFF_BS];
    __builtin_amdgcn_sched_barrier(0);

    // ---- colour sine layer (MFMA 64x128) ----
    f32x16 accc[2];
    zero_acc(accc);
    mfma_layer<2, 4>(Wcs, 32 * LDC, 1, h, accc);
    float hc[2][16];
    film_act<HW, false, 2>(accc, sm + OFF_GC, sm + OFF_CC, hf, hc, hc);

    // ---- colour linear (MFMA 32x64) ----
    f32x16 accf[1];
    zero_acc(accf);
    mfma_layer<1, 2>(Wfs, 0, 1, hc, accf);

    if (valid) {
      float* fo = a.feat + gp * CF + 4 * hf;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 v;
        v.x = accf[0][4 * g + 0] + sm[OFF_BF + 8 * g + 4 * hf + 0];
        v.y = accf[0][4 * g + 1] + sm[OFF_BF + 8 * g + 4 * hf + 1];
        v.z = accf[0][4 * g + 2] + sm[OFF_BF + 8 * g + 4 * hf + 2];
        v.w = accf[0][4 * g + 3] + sm[OFF_BF + 8 * g + 4 * hf + 3];
        *reinterpret_cast<float4*>(fo + 8 * g) = v;
      }
      if (hf == 0) a.sigma[gp] = sig;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ------------------------------------------------------------------------------------
// backward "data" pass
// ------------------------------------------------------------------------------------
struct BwdArgs {
  cips_siren_weights w;
  const float* points;
  const float* dfeat;
  const float* dsigma;
  unsigned short *h1h, *h1l, *h2h, *h2l, *hch, *hcl, *da2h, *da2l, *dach, *dacl;
  float* red;
  int B, P, chunk, chunks;
};

// Transpose-reduce 32 registers across the 32 lanes of each wave half:
// on return v[0] of lane j (j = lane & 31) holds sum over the half's lanes of the input v[j].
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

__device__ __forceinline__ unsigned short f2bf_rne(float v) {
  unsigned u = __float_as_uint(v);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
// store Q*16 per-lane features of layout "N" into [rows][Q*32] row-major split-bf16 planes (x = hi + lo):
// the k-major operands of the bf16x3 weight-gradient GEMMs (cips_gemm_bf16x3_km), 8 bytes per plane per store
template <int Q>
__device__ __forceinline__ void store_layoutN(unsigned short* base_hi, unsigned short* base_lo, long long row, int hf,
                                              const float (&v)[Q][16]) {
  unsigned short* ph = base_hi + row * (Q * 32) + 4 * hf;
  unsigned short* pl = base_lo + row * (Q * 32) + 4 * hf;
#pragma unroll
  for (int q = 0; q < Q; ++q)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      unsigned short h[4], l[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x = v[q][4 * g + e];
        h[e] = f2bf_rne(x);
        l[e] = f2bf_rne(x - __uint_as_float(((unsigned)h[e]) << 16));
      }
      *reinterpret_cast<uint2*>(ph + q * 32 + 8 * g) = make_uint2(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16));
      *reinterpret_cast<uint2*>(pl + q * 32 + 8 * g) = make_uint2(l[0] | ((unsigned)l[1] << 16), l[2] | ((unsigned)l[3] << 16));
    }
}

// F32OUT (round 6, the all-fp32 leg): the same values stored as fp32 rows [point][Q * 32] through the hi pointers (the lo pointers
// are unused) — the operands of fp32-MFMA weight-gradient GEMMs, so that no split operand takes part anywhere in that leg
template <int Q, bool F32OUT>
__device__ __forceinline__ void store_act(unsigned short* base_hi, unsigned short* base_lo, long long row, int hf, const float (&v)[Q][16]) {
  if constexpr (F32OUT) {
    float* pf = reinterpret_cast<float*>(base_hi) + row * (Q * 32) + 4 * hf;
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(pf + q * 32 + 8 * g) = make_float4(v[q][4 * g], v[q][4 * g + 1], v[q][4 * g + 2], v[q][4 * g + 3]);
  } else {
    store_layoutN<Q>(base_hi, base_lo, row, hf, v);
  }
}

template <bool HW, bool F32OUT = false>
__global__ __launch_bounds__(256, 1) void siren_bwd_kernel(BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int b = blockIdx.y;
  stage_weights(sm, a.w, b);
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int l31 = lane & 31, hf = lane >> 5;
  const int cstart = blockIdx.x * a.chunk;
  const int cend = min(cstart + a.chunk, a.P);
  const float* W1s = sm + OFF_W1 + l31 * LD1 + 4 * hf;   // forward:  row = out feature (lane)
  const float* Wcs = sm + OFF_WC + l31 * LDC + 4 * hf;
  const float* W1t = sm + OFF_W1 + 4 * hf * LD1 + l31;   // backward: row = k feature, col = lane
  const float* Wct = sm + OFF_WC + 4 * hf * LDC + l31;
  const float* Wft = sm + OFF_WF + 4 * hf * LDF + l31;
  const float4* L0 = reinterpret_cast<const float4*>(sm + OFF_L0);

  // running per-wave reductions (lane j of each half: register-index j of the group)
  float r_da1[2] = {0.f, 0.f}, r_x[3][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
  float r_da2[2] = {0.f, 0.f}, r_dac = 0.f, r_ws[2] = {0.f, 0.f}, r_df = 0.f;

  for (int p0 = cstart + wave * 32; p0 < cend; p0 += nwaves * 32) {
    const int p = p0 + l31;
    const bool valid = p < cend;
    const long long gp = (long long)b * a.P + (valid ? p : cend - 1);
    const float px = a.points[gp * 3 + 0], py = a.points[gp * 3 + 1], pz = a.points[gp * 3 + 2];
    const float dsg = valid ? a.dsigma[gp] : 0.f;

    // ---- recompute layer 0 ----
    float h[4][16];
    layer0<HW, false>(L0, hf, px, py, pz, h, h);   // cos of layer 0 is recomputed at the end (saves 64 live VGPRs)
    if (valid) store_act<4, F32OUT>(a.h1h, a.h1l, gp, hf, h);

    // ---- recompute layer 1 ----
    f32x16 acc[4];
    zero_acc(acc);
    mfma_layer<4, 4>(W1s, 32 * LD1, 1, h, acc);
    float cs2[4][16];
    film_act<HW, true, 4>(acc, sm + OFF_G1, sm + OFF_C1, hf, h, cs2);
    if (valid) store_act<4, F32OUT>(a.h2h, a.h2l, gp, hf, h);
    // sum_p dsigma * h2  (gradient of final_layer.weight)
#pragma unroll
    for (int gI = 0; gI < 2; ++gI) {
      float v[32];
#pragma unroll
      for (int i = 0; i < 32; ++i) v[i] = dsg * h[2 * gI + (i >> 4)][i & 15];
      r_ws[gI] += reduce32(v, lane);
    }

    // ---- recompute colour sine layer ----
    f32x16 accc[2];
    zero_acc(accc);
    mfma_layer<2, 4>(Wcs, 32 * LDC, 1, h, accc);
    float hc[2][16], csc[2][16];
    film_act<HW, true, 2>(accc, sm + OFF_GC, sm + OFF_CC, hf, hc, csc);
    if (valid) store_act<2, F32OUT>(a.hch, a.hcl, gp, hf, hc);

    // ---- d hc = Wf^T dfeat   (K = 32 channels, M = 64) ----
    float df[1][16];
    {
      const float* dp = a.dfeat + gp * CF + 4 * hf;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 v = valid ? *reinterpret_cast<const float4*>(dp + 8 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
        df[0][4 * g + 0] = v.x; df[0][4 * g + 1] = v.y; df[0][4 * g + 2] = v.z; df[0][4 * g + 3] = v.w;
      }
    }
    {
      float v[32];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = df[0][i];
      v[16] = (hf == 0) ? dsg : 0.f;
#pragma unroll
      for (int i = 17; i < 32; ++i) v[i] = 0.f;
      r_df += reduce32(v, lane);
    }
    zero_acc(accc);
    mfma_layer<2, 1>(Wft, 32, LDF, df, accc);
    // dac = d hc * cos ; dprec = gc * dac
    float dpc[2][16];
    {
      float v[32];
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float d = accc[q][r] * csc[q][r];
          hc[q][r] = d;  // reuse as dac
          v[q * 16 + r] = d;
          dpc[q][r] = sm[OFF_GC + featidx(q, r, 0) + 4 * hf] * d;
          if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
      if (valid) store_act<2, F32OUT>(a.dach, a.dacl, gp, hf, hc);
      r_dac += reduce32(v, lane);
    }

    // ---- d h2 = Wc^T dprec + ws * dsigma   (K = 64, M = 128) ----
    zero_acc(acc);
    mfma_layer<4, 2>(Wct, 32, LDC, dpc, acc);
    float dp2[4][16];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int f = featidx(q, r, 0) + 4 * hf;
        float d = fmaf(sm[OFF_WS + f], dsg, acc[q][r]) * cs2[q][r];
        h[q][r] = d;  // reuse as da2
        dp2[q][r] = sm[OFF_G1 + f] * d;
        if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
    if (valid) store_act<4, F32OUT>(a.da2h, a.da2l, gp, hf, h);
#pragma unroll
    for (int gI = 0; gI < 2; ++gI) {
      float v[32];
#pragma unroll
      for (int i = 0; i < 32; ++i) v[i] = h[2 * gI + (i >> 4)][i & 15];
      r_da2[gI] += reduce32(v, lane);
    }

    // ---- d h1 = W1^T dpre2   (K = 128, M = 128) ----
    zero_acc(acc);
    mfma_layer<4, 4>(W1t, 32, LD1, dp2, acc);
#pragma unroll
    for (int gI = 0; gI < 2; ++gI) {
      float v[32], t[32];
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        float4 pk = L0[featidx(2 * gI + (i >> 4), i & 15, 0) + 4 * hf];
        float sn, cs;
        film_sincos<HW>(fmaf(pk.x, px, fmaf(pk.y, py, fmaf(pk.z, pz, pk.w))), &sn, &cs);
        v[i] = acc[2 * gI + (i >> 4)][i & 15] * cs;
        t[i] = v[i];
        if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
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
  }

  // ---- write this wave's partial reductions ----
  float* row = a.red + ((long long)(b * a.chunks + blockIdx.x) * nwaves + wave) * RED_W;
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
  if (l31 < 16) row[832 + mfma_row(l31, hf)] = r_df;        // sum_p dfeat[ch]
  if (l31 == 16 && hf == 0) row[864] = r_df;                 // sum_p dsigma
  if (l31 >= 17 && l31 < 20 && hf == 0) row[864 + (l31 - 16)] = 0.f;
}

constexpr int FWD_CHUNK = 2048;
constexpr int BWD_CHUNK = 2048;

}  // namespace

extern "C" int cips_siren_fwd(const cips_siren_weights* w, const float* points, float* feat,
                              float* sigma, int B, int P, cips_stream_t stream) {
  if (!w || B <= 0 || P <= 0) return (int)hipErrorInvalidValue;
  FwdArgs a;
  a.w = *w; a.points = points; a.feat = feat; a.sigma = sigma; a.B = B; a.P = P;
  a.chunk = FWD_CHUNK;
  dim3 grid((P + a.chunk - 1) / a.chunk, B);
  size_t smem = SMEM_FLOATS * sizeof(float);
  static bool attr_set = false;
  CIPS_PER_DEVICE(attr_set, false);
  if (!attr_set) {
    hipFuncSetAttribute((const void*)siren_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipFuncSetAttribute((const void*)siren_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  if ((w->trig_mode & 1))
    hipLaunchKernelGGL(siren_fwd_kernel<true>, grid, dim3(512), smem, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(siren_fwd_kernel<false>, grid, dim3(512), smem, (hipStream_t)stream, a);
  return CIPS_CHECK_LAUNCH();
}

extern "C" int cips_siren_bwd_rows(int B, int P) {
  int chunks = (P + BWD_CHUNK - 1) / BWD_CHUNK;
  return B * chunks * 4;
}

static int siren_bwd_data_launch(const cips_siren_weights* w, const float* points, const float* dfeat, const float* dsigma, void* h1_hi,
                                 void* h1_lo, void* h2_hi, void* h2_lo, void* hc_hi, void* hc_lo, void* da2_hi, void* da2_lo, void* dac_hi,
                                 void* dac_lo, float* red, int B, int P, bool f32out, cips_stream_t stream);

extern "C" int cips_siren_bwd_data(const cips_siren_weights* w, const float* points,
                                   const float* dfeat, const float* dsigma, void* h1_hi, void* h1_lo, void* h2_hi,
                                   void* h2_lo, void* hc_hi, void* hc_lo, void* da2_hi, void* da2_lo, void* dac_hi,
                                   void* dac_lo, float* red, int B, int P, cips_stream_t stream) {
  return siren_bwd_data_launch(w, points, dfeat, dsigma, h1_hi, h1_lo, h2_hi, h2_lo, hc_hi, hc_lo, da2_hi, da2_lo, dac_hi, dac_lo, red, B, P,
                               false, stream);
}

extern "C" int cips_siren_bwd_data_f32(const cips_siren_weights* w, const float* points, const float* dfeat, const float* dsigma, float* h1,
                                       float* h2, float* hc, float* da2, float* dac, float* red, int B, int P, cips_stream_t stream) {
  if (!h1 || !h2 || !hc || !da2 || !dac) return (int)hipErrorInvalidValue;
  return siren_bwd_data_launch(w, points, dfeat, dsigma, h1, nullptr, h2, nullptr, hc, nullptr, da2, nullptr, dac, nullptr, red, B, P, true, stream);
}

static int siren_bwd_data_launch(const cips_siren_weights* w, const float* points, const float* dfeat, const float* dsigma, void* h1_hi,
                                 void* h1_lo, void* h2_hi, void* h2_lo, void* hc_hi, void* hc_lo, void* da2_hi, void* da2_lo, void* dac_hi,
                                 void* dac_lo, float* red, int B, int P, bool f32out, cips_stream_t stream) {
  if (!w || B <= 0 || P <= 0) return (int)hipErrorInvalidValue;
  BwdArgs a;
  a.w = *w; a.points = points; a.dfeat = dfeat; a.dsigma = dsigma;
  a.h1h = (unsigned short*)h1_hi; a.h1l = (unsigned short*)h1_lo; a.h2h = (unsigned short*)h2_hi; a.h2l = (unsigned short*)h2_lo;
  a.hch = (unsigned short*)hc_hi; a.hcl = (unsigned short*)hc_lo; a.da2h = (unsigned short*)da2_hi; a.da2l = (unsigned short*)da2_lo;
  a.dach = (unsigned short*)dac_hi; a.dacl = (unsigned short*)dac_lo; a.red = red;
  a.B = B; a.P = P; a.chunk = BWD_CHUNK; a.chunks = (P + BWD_CHUNK - 1) / BWD_CHUNK;
  dim3 grid(a.chunks, B);
  size_t smem = SMEM_FLOATS * sizeof(float);
  static bool attr_set = false;
  CIPS_PER_DEVICE(attr_set, false);
  if (!attr_set) {
    hipFuncSetAttribute((const void*)siren_bwd_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipFuncSetAttribute((const void*)siren_bwd_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipFuncSetAttribute((const void*)siren_bwd_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipFuncSetAttribute((const void*)siren_bwd_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  const bool hw = (w->trig_mode & 1);
  if (hw && f32out) hipLaunchKernelGGL((siren_bwd_kernel<true, true>), grid, dim3(256), smem, (hipStream_t)stream, a);
  else if (hw) hipLaunchKernelGGL((siren_bwd_kernel<true, false>), grid, dim3(256), smem, (hipStream_t)stream, a);
  else if (f32out) hipLaunchKernelGGL((siren_bwd_kernel<false, true>), grid, dim3(256), smem, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((siren_bwd_kernel<false, false>), grid, dim3(256), smem, (hipStream_t)stream, a);
  return CIPS_CHECK_LAUNCH();
}
