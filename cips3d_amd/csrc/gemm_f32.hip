// gemm_f32.hip — batched fp32 GEMM on v_mfma_f32_32x32x2_f32 (exact fp32, 157 TFLOP/s
// peak on MI355X) with the epilogues the CIPS-3D hot path needs fused in.
//
// Replaces, on the hot path: torch.bmm in SinStyleMod.forward_bmm
// (exp/comm/models/mod_conv_fc.py:489) + the LeakyReLU / skip-add that follow it
// (exp/cips3d/models/generator.py:949-974), their autograd backward GEMMs, the SIREN
// weight-gradient contractions and the discriminator's im2col GEMMs.
//
// Tiling (CDNA4, wave64): 128x128x32 workgroup tile, 256 threads = 4 waves in a 2x2
// grid, each wave owns a 64x64 sub-tile = 2x2 MFMA tiles of 32x32 (64 accumulator
// VGPRs).  Operands are staged HBM -> VGPR -> LDS with a one-tile register prefetch:
// the global loads for tile t+1 are in flight while tile t is multiplied out of LDS.
// LDS images are chosen so that every ds_read_b32 of an MFMA operand is bank-conflict
// free: A row-major is stored [m][33] (odd stride), A k-major and B are stored [k][128].
// fp32 MFMA issues one instruction per 64 cycles per SIMD, so one ds_read_b32 per operand
// per MFMA is ~3% of LDS bandwidth: staging is never the limiter, MFMA issue is.
// Workgroup ids are remapped so that each XCD (block id % 8) walks a contiguous run of
// tiles with the N index fastest: the N-tiles that share an A panel hit the same L2.
#include "common.h"
#include "../../include/cips3d_hip.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDA_M = BK + 1;   // A row-major image  [BM][33]
constexpr int LDA_K = BM;       // A k-major image    [BK][128]
constexpr int LDB = BN;         // B image            [BK][128]
constexpr int SMEM_A = BM * LDA_M;  // >= BK*LDA_K
constexpr int LDB_N = BK + 1;      // B n-major image    [BN][33]
constexpr int SMEM_B = BN * LDB_N;  // >= BK*LDB

struct Args {
  cips_gemm_desc d;
  int tiles_m, tiles_n, total;
};

template <bool A_KMAJOR, bool B_NMAJOR>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(Args g) {
  __shared__ __attribute__((aligned(16))) float smem[SMEM_A + SMEM_B];
  float* As = smem;
  float* Bs = smem + SMEM_A;
  const cips_gemm_desc& d = g.d;

  // ---- XCD-aware bijective remap of the workgroup id (block b runs on XCD b % 8) ----
  int bid = blockIdx.x;
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

  const float* __restrict__ A = d.A + (long long)bz * d.strideA;
  const float* __restrict__ B = d.B + (long long)bz * d.strideB;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hf = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[4], rb[4];
  const int M = d.M, N = d.N, K = d.K;

  auto load_tile = [&](int k0) {
    if (!A_KMAJOR) {
      // A (M,K) row-major: thread -> rows tid/8 + 32*i, 4 consecutive k at (tid%8)*4
      const int c4 = (tid & 7) * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int row = (tid >> 3) + 32 * i;
        int gm = m0 + row, gk = k0 + c4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gm < M && gk < K) v = *reinterpret_cast<const float4*>(A + (long long)gm * d.lda + gk);
        ra[i] = v;
      }
    } else {
      // A (K,M) row-major: thread -> k rows tid/32 + 8*i, 4 consecutive m at (tid%32)*4
      const int c4 = (tid & 31) * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int kr = (tid >> 5) + 8 * i;
        int gk = k0 + kr, gm = m0 + c4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gk < K && gm < M) v = *reinterpret_cast<const float4*>(A + (long long)gk * d.lda + gm);
        ra[i] = v;
      }
    }
    if (!B_NMAJOR) {
      const int c4 = (tid & 31) * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int kr = (tid >> 5) + 8 * i;
        int gk = k0 + kr, gn = n0 + c4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gk < K && gn < N) v = *reinterpret_cast<const float4*>(B + (long long)gk * d.ldb + gn);
        rb[i] = v;
      }
    } else {
      // B (N,K) row-major: thread -> rows n = tid/8 + 32*i, 4 consecutive k at (tid%8)*4
      const int c4 = (tid & 7) * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int row = (tid >> 3) + 32 * i;
        int gn = n0 + row, gk = k0 + c4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gn < N && gk < K) v = *reinterpret_cast<const float4*>(B + (long long)gn * d.ldb + gk);
        rb[i] = v;
      }
    }
  };

  auto store_tile = [&]() {
    if (!A_KMAJOR) {
      const int c4 = (tid & 7) * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int row = (tid >> 3) + 32 * i;
        float* p = As + row * LDA_M + c4;
        p[0] = ra[i].x; p[1] = ra[i].y; p[2] = ra[i].z; p[3] = ra[i].w;
      }
    } else {
      const int c4 = (tid & 31) * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int kr = (tid >> 5) + 8 * i;
        *reinterpret_cast<float4*>(As + kr * LDA_K + c4) = ra[i];
      }
    }
    if (!B_NMAJOR) {
      const int c4 = (tid & 31) * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int kr = (tid >> 5) + 8 * i;
        *reinterpret_cast<float4*>(Bs + kr * LDB + c4) = rb[i];
      }
    } else {
      const int c4 = (tid & 7) * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int row = (tid >> 3) + 32 * i;
        float* p = Bs + row * LDB_N + c4;
        p[0] = rb[i].x; p[1] = rb[i].y; p[2] = rb[i].z; p[3] = rb[i].w;
      }
    }
  };

  const int nk = (K + BK - 1) / BK;
  load_tile(0);
  store_tile();
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) load_tile((kt + 1) * BK);
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      float a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        int m = wm * 64 + i * 32 + l31;
        a[i] = A_KMAJOR ? As[(kk * 2 + hf) * LDA_K + m] : As[m * LDA_M + kk * 2 + hf];
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int n = wn * 64 + j * 32 + l31;
        b[j] = B_NMAJOR ? Bs[n * LDB_N + kk * 2 + hf] : Bs[(kk * 2 + hf) * LDB + n];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
    if (kt + 1 < nk) {
      store_tile();
      __syncthreads();
    }
  }

  // ---------------- epilogue ----------------
  const long long cbase = (long long)bz * d.strideC;
  const float alpha = (d.alpha == 0.f) ? 1.f : d.alpha;
  const float gain = (d.act == 2) ? d.act_gain : 1.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + l31;
      if (col >= N) continue;
      float bias = d.bias ? d.bias[col] : 0.f;
      float rw0 = 0.f, rw1 = 0.f, rw2 = 0.f;
      if (d.rgb_g) { rw0 = d.rgb_w[col]; rw1 = d.rgb_w[N + col]; rw2 = d.rgb_w[2 * N + col]; }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + mfma_row(r, hf);
        if (row >= M) continue;
        const long long off = cbase + (long long)row * d.ldc + col;
        float v = acc[i][j][r] * alpha + bias;
        if (d.bias_m) v += d.bias_m[row];
        if (d.add) v += d.add[off];
        if (d.rgb_g) {
          const float* gp = d.rgb_g + ((long long)bz * M + row) * 3;
          v = fmaf(gp[0], rw0, fmaf(gp[1], rw1, fmaf(gp[2], rw2, v)));
        }
        if (d.C_unmasked) d.C_unmasked[off] = v;
        if (d.mask) v *= (d.mask[off] > 0.f ? 1.f : d.slope) * gain;
        if (d.act) v = lrelu(v, d.slope) * gain;
        d.C[off] = v;
        if (d.C2) d.C2[off] = v + (d.resid ? d.resid[off] : 0.f);
      }
    }
  }
}

}  // namespace

extern "C" int cips_gemm_f32(const cips_gemm_desc* d, cips_stream_t stream) {
  if (!d || d->M <= 0 || d->N <= 0 || d->K <= 0 || d->batch <= 0) return (int)hipErrorInvalidValue;
  // vector (16 B) global loads: leading dims and the contiguous extents must be multiples of 4 floats
  if ((d->lda & 3) || (d->ldb & 3)) return (int)hipErrorInvalidValue;
  if (d->a_kmajor ? (d->M & 3) : (d->K & 3)) return (int)hipErrorInvalidValue;
  if (d->b_nmajor ? (d->K & 3) : (d->N & 3)) return (int)hipErrorInvalidValue;
  if ((d->strideA & 3) || (d->strideB & 3)) return (int)hipErrorInvalidValue;
  Args g;
  g.d = *d;
  g.tiles_m = (d->M + BM - 1) / BM;
  g.tiles_n = (d->N + BN - 1) / BN;
  long long total = (long long)g.tiles_m * g.tiles_n * d->batch;
  if (total > 0x7fffffffLL) return (int)hipErrorInvalidValue;
  g.total = (int)total;
  hipStream_t s = (hipStream_t)stream;
  const int variant = (d->a_kmajor ? 1 : 0) | (d->b_nmajor ? 2 : 0);
  switch (variant) {
    case 0: hipLaunchKernelGGL((gemm_f32_kernel<false, false>), dim3(g.total), dim3(256), 0, s, g); break;
    case 1: hipLaunchKernelGGL((gemm_f32_kernel<true, false>), dim3(g.total), dim3(256), 0, s, g); break;
    case 2: hipLaunchKernelGGL((gemm_f32_kernel<false, true>), dim3(g.total), dim3(256), 0, s, g); break;
    default: hipLaunchKernelGGL((gemm_f32_kernel<true, true>), dim3(g.total), dim3(256), 0, s, g); break;
  }
  return CIPS_CHECK_LAUNCH();
}
