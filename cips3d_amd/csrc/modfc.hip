// modfc.hip — helpers around the CIPS INR head's modulated FC layers (H4) for gfx950.
//
// The reference (exp/comm/models/mod_conv_fc.py:470-489, SinStyleMod.forward_bmm) builds a
// per-image weight  Wb = W * (s+1)[:,None] * rsqrt(sum_in (W*(s+1))^2 + eps)[None,:]  and runs
// torch.bmm(x, Wb).  Here `cips_modfc_prep` produces Wb (and its transpose for the dX GEMM)
// once per layer — B x 1 MB, L2/MALL resident — and the bmm itself is cips_gemm_f32 with the
// LeakyReLU / skip epilogue fused.  ToRGB (generator.py:983-1006) is a 512 -> 3 projection:
// pure HBM streaming, one wave per pixel row.
#include "common.h"
#include <cstdlib>
#include "../../include/cips3d_hip.h"

namespace {

// ---- prep: demod[b][n] and wb / wbt ------------------------------------------------
// grid (out/32, B), 256 threads = 32 columns x 8 k-groups
__global__ __launch_bounds__(256) void modfc_prep_kernel(const float* __restrict__ W, const float* __restrict__ s,
                                                         float* __restrict__ wb, float* __restrict__ wbt,
                                                         float* __restrict__ demod, int in_dim, int out_dim, float eps) {
  __shared__ float red[8][33];
  __shared__ float dsh[32];
  const int b = blockIdx.y;
  const int c = threadIdx.x & 31, kg = threadIdx.x >> 5;
  const int n = blockIdx.x * 32 + c;
  const float* sb = s + (long long)b * in_dim;
  float q = 0.f;
  if (n < out_dim)
    for (int k = kg; k < in_dim; k += 8) {
      float u = W[(long long)k * out_dim + n] * (sb[k] + 1.f);
      q = fmaf(u, u, q);
    }
  red[kg][c] = q;
  __syncthreads();
  if (kg == 0) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) t += red[g][c];
    float d = rsqrtf(t + eps);
    dsh[c] = d;
    if (n < out_dim) demod[(long long)b * out_dim + n] = d;
  }
  __syncthreads();
  float* wbb = wb + (long long)b * in_dim * out_dim;
  float* wtb = wbt + (long long)b * in_dim * out_dim;
  if (n < out_dim) {
    const float d = dsh[c];
    for (int k = kg; k < in_dim; k += 8)
      wbb[(long long)k * out_dim + n] = W[(long long)k * out_dim + n] * (sb[k] + 1.f) * d;
  }
  // transposed copy: consecutive threads walk k (contiguous in wbt)
  for (int idx = threadIdx.x; idx < 32 * in_dim; idx += 256) {
    const int k = idx % in_dim, cc = idx / in_dim;
    const int nn = blockIdx.x * 32 + cc;
    if (nn < out_dim)
      wtb[(long long)nn * in_dim + k] = W[(long long)k * out_dim + nn] * (sb[k] + 1.f) * dsh[cc];
  }
}

// bf16x3 operand planes of Wb [in][out] and Wbt [out][in] (hi/lo each), in two launches:
//   (1) demod[b][n] = rsqrt(sum_k (W[k][n] (s[b][k]+1))^2 + eps)          grid (out/32, B)
//   (2) 32x32 tiles of v = W (s+1) demod, split to hi/lo, written row-major and (through LDS) transposed,
//       both with coalesced accesses                                            grid (out/32, in/32, B)
__device__ __forceinline__ unsigned short f2bf_rne(float v) {
  unsigned u = __float_as_uint(v);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__global__ __launch_bounds__(256) void modfc_demod_kernel(const float* __restrict__ W, const float* __restrict__ s,
                                                          float* __restrict__ demod, int in_dim, int out_dim, float eps) {
  __shared__ float red[8][33];
  const int b = blockIdx.y;
  const int c = threadIdx.x & 31, kg = threadIdx.x >> 5;
  const int n = blockIdx.x * 32 + c;
  const float* sb = s + (long long)b * in_dim;
  float q = 0.f;
  if (n < out_dim)
    for (int k = kg; k < in_dim; k += 8) {
      float u = W[(long long)k * out_dim + n] * (sb[k] + 1.f);
      q = fmaf(u, u, q);
    }
  red[kg][c] = q;
  __syncthreads();
  if (kg == 0 && n < out_dim) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) t += red[g][c];
    demod[(long long)b * out_dim + n] = rsqrtf(t + eps);
  }
}
__global__ __launch_bounds__(256) void modfc_planes_kernel(const float* __restrict__ W, const float* __restrict__ s,
                                                           const float* __restrict__ demod,
                                                           unsigned short* __restrict__ wbh, unsigned short* __restrict__ wbl,
                                                           unsigned short* __restrict__ wth, unsigned short* __restrict__ wtl,
                                                           int in_dim, int out_dim) {
  __shared__ unsigned short th[32][33], tl[32][33];
  const int b = blockIdx.z;
  const int n0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const long long base = (long long)b * in_dim * out_dim;
  for (int kk = ty; kk < 32; kk += 8) {
    const int k = k0 + kk, n = n0 + tx;
    unsigned short h = 0, l = 0;
    if (k < in_dim && n < out_dim) {
      const float v = W[(long long)k * out_dim + n] * (s[(long long)b * in_dim + k] + 1.f) * demod[(long long)b * out_dim + n];
      h = f2bf_rne(v);
      l = f2bf_rne(v - __uint_as_float(((unsigned)h) << 16));
      wbh[base + (long long)k * out_dim + n] = h;
      wbl[base + (long long)k * out_dim + n] = l;
    }
    th[kk][tx] = h; tl[kk][tx] = l;
  }
  __syncthreads();
  for (int nn = ty; nn < 32; nn += 8) {
    const int n = n0 + nn, k = k0 + tx;
    if (k < in_dim && n < out_dim) {
      wth[base + (long long)n * in_dim + k] = th[tx][nn];
      wtl[base + (long long)n * in_dim + k] = tl[tx][nn];
    }
  }
}

// ---- prep backward -------------------------------------------------------------------
// u = W*(s+1), q_n = sum_k u^2 + eps, d = q^-1/2, Wb = u*d.  With G = dL/dWb:
//   c_n  = sum_k G_kn u_kn
//   du   = d_n * (G_kn - d_n^2 u_kn c_n)
//   dW_kn = sum_b (s_bk+1) du_bkn ;  ds_bk = sum_n W_kn du_bkn
__global__ __launch_bounds__(256) void modfc_prep_bwd_c_kernel(const float* __restrict__ W, const float* __restrict__ s,
                                                               const float* __restrict__ G, float* __restrict__ cbuf,
                                                               int in_dim, int out_dim) {
  __shared__ float red[8][33];
  const int b = blockIdx.y;
  const int c = threadIdx.x & 31, kg = threadIdx.x >> 5;
  const int n = blockIdx.x * 32 + c;
  const float* sb = s + (long long)b * in_dim;
  const float* Gb = G + (long long)b * in_dim * out_dim;
  float q = 0.f;
  if (n < out_dim)
    for (int k = kg; k < in_dim; k += 8)
      q = fmaf(Gb[(long long)k * out_dim + n], W[(long long)k * out_dim + n] * (sb[k] + 1.f), q);
  red[kg][c] = q;
  __syncthreads();
  if (kg == 0 && n < out_dim) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) t += red[g][c];
    cbuf[(long long)b * out_dim + n] = t;
  }
}

__global__ __launch_bounds__(256) void modfc_prep_bwd_w_kernel(const float* __restrict__ W, const float* __restrict__ s,
                                                               const float* __restrict__ demod, const float* __restrict__ G,
                                                               const float* __restrict__ cbuf, float* __restrict__ dW,
                                                               int B, int in_dim, int out_dim) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)in_dim * out_dim) return;
  const int k = (int)(idx / out_dim), n = (int)(idx % out_dim);
  const float w = W[idx];
  float acc = 0.f;
  for (int b = 0; b < B; ++b) {   // fixed order: deterministic
    const float m = s[(long long)b * in_dim + k] + 1.f;
    const float d = demod[(long long)b * out_dim + n];
    const float g = G[((long long)b * in_dim + k) * out_dim + n];
    const float du = d * (g - d * d * (w * m) * cbuf[(long long)b * out_dim + n]);
    acc = fmaf(m, du, acc);
  }
  dW[idx] = acc;
}

// one wave per (b, k): lanes stride over n
__global__ __launch_bounds__(256) void modfc_prep_bwd_s_kernel(const float* __restrict__ W, const float* __restrict__ s,
                                                               const float* __restrict__ demod, const float* __restrict__ G,
                                                               const float* __restrict__ cbuf, float* __restrict__ ds,
                                                               int B, int in_dim, int out_dim) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long long)B * in_dim) return;
  const int b = (int)(row / in_dim), k = (int)(row % in_dim);
  const float m = s[row] + 1.f;
  float acc = 0.f;
  for (int n = lane; n < out_dim; n += 64) {
    const float w = W[(long long)k * out_dim + n];
    const float d = demod[(long long)b * out_dim + n];
    const float g = G[row * out_dim + n];
    const float du = d * (g - d * d * (w * m) * cbuf[(long long)b * out_dim + n]);
    acc = fmaf(w, du, acc);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
  if (lane == 0) ds[row] = acc;
}

// ---- ToRGB --------------------------------------------------------------------------
typedef unsigned short u16;
__device__ __forceinline__ float bf2f(u16 h) { return __uint_as_float(((unsigned)h) << 16); }
// load 4 consecutive activations either from fp32 or from bf16 hi/lo planes
template <bool X3>
__device__ __forceinline__ float4 ldx4(const void* xa, const void* xb, long long e) {
  if (!X3) return *reinterpret_cast<const float4*>((const float*)xa + e);
  const uint2 h = *reinterpret_cast<const uint2*>((const u16*)xa + e);
  const uint2 l = *reinterpret_cast<const uint2*>((const u16*)xb + e);
  float4 v;
  v.x = __uint_as_float(h.x << 16) + __uint_as_float(l.x << 16);
  v.y = __uint_as_float(h.x & 0xffff0000u) + __uint_as_float(l.x & 0xffff0000u);
  v.z = __uint_as_float(h.y << 16) + __uint_as_float(l.y << 16);
  v.w = __uint_as_float(h.y & 0xffff0000u) + __uint_as_float(l.y & 0xffff0000u);
  return v;
}

// rgb[m][c] (+)= sum_k x[m][k] w[c][k] + bias[c];  one wave per row, float4 lanes
template <bool X3>
__global__ __launch_bounds__(256) void torgb_fwd_kernel(const void* __restrict__ xa, const void* __restrict__ xb,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ rgb,
                                                        long long M, int K, int accumulate) {
  const int lane = threadIdx.x & 63;
  const long long wave0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long nwaves = (long long)gridDim.x * 4;
  for (long long m = wave0; m < M; m += nwaves) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
      const float4 v = ldx4<X3>(xa, xb, m * K + k);
      const float4 w0 = *reinterpret_cast<const float4*>(w + k);
      const float4 w1 = *reinterpret_cast<const float4*>(w + K + k);
      const float4 w2 = *reinterpret_cast<const float4*>(w + 2 * K + k);
      a0 += v.x * w0.x + v.y * w0.y + v.z * w0.z + v.w * w0.w;
      a1 += v.x * w1.x + v.y * w1.y + v.z * w1.z + v.w * w1.w;
      a2 += v.x * w2.x + v.y * w2.y + v.z * w2.z + v.w * w2.w;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      a0 += __shfl_xor(a0, off); a1 += __shfl_xor(a1, off); a2 += __shfl_xor(a2, off);
    }
    if (lane < 3) {
      float v = (lane == 0) ? a0 : (lane == 1 ? a1 : a2);
      v += bias ? bias[lane] : 0.f;
      float* o = rgb + m * 3 + lane;
      *o = accumulate ? (*o + v) : v;
    }
  }
}

// Split-plane form for K = 512 (the head's width): a lane owns 8 consecutive k (one 16-byte load per plane and row),
// the three weight rows of its k-range stay in registers for all rows of the wave, four rows are in flight per iteration.
// (The generic kernel above re-reads the weights per row and moves 8 bytes per load: 79 us = 3.4 TB/s at C2.)
__device__ __forceinline__ void ld8x3(const u16* hi, const u16* lo, long long e, float (&v)[8]) {
  const uint4 h = *reinterpret_cast<const uint4*>(hi + e);
  const uint4 l = *reinterpret_cast<const uint4*>(lo + e);
  const unsigned hw[4] = {h.x, h.y, h.z, h.w}, lw[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    v[2 * q] = __uint_as_float(hw[q] << 16) + __uint_as_float(lw[q] << 16);
    v[2 * q + 1] = __uint_as_float(hw[q] & 0xffff0000u) + __uint_as_float(lw[q] & 0xffff0000u);
  }
}

__global__ __launch_bounds__(256) void torgb_fwd_x3_k512_kernel(const u16* __restrict__ xh, const u16* __restrict__ xl,
                                                                const float* __restrict__ w, const float* __restrict__ bias,
                                                                float* __restrict__ rgb, long long M, int accumulate) {
  constexpr int K = 512, R = 4;
  const int lane = threadIdx.x & 63;
  const long long wave0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long nwaves = (long long)gridDim.x * 4;
  float wr[3][8];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int q = 0; q < 8; ++q) wr[c][q] = w[c * K + lane * 8 + q];
  const float bv = (bias && lane < 3) ? bias[lane] : 0.f;
  for (long long m0 = wave0 * R; m0 < M; m0 += nwaves * R) {
    float a[R][3];
    float v[R][8];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const long long m = (m0 + r < M) ? m0 + r : M - 1;
      ld8x3(xh, xl, m * K + lane * 8, v[r]);
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) t = fmaf(v[r][q], wr[c][q], t);
        a[r][c] = t;
      }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) a[r][c] += __shfl_xor(a[r][c], off);
    if (lane < 3) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (m0 + r < M) {
          const float t = ((lane == 0) ? a[r][0] : (lane == 1 ? a[r][1] : a[r][2])) + bv;
          float* o = rgb + (m0 + r) * 3 + lane;
          *o = accumulate ? (*o + t) : t;
        }
      }
    }
  }
}

constexpr int TORGB_ROWS = 128;  // rows per partial chunk

// Same for the weight gradient: partial[chunk][c][k] = sum over the chunk's 128 rows of drgb[m][c] x[m][k], K = 512.
// 64 lanes x 8 k cover a row; the four waves take rows m0 + wave, + 4, ... (32 each), combined in wave order via LDS.
constexpr int TORGB_MAXJOBS = 8;
struct TorgbJobs { const u16* xh[TORGB_MAXJOBS]; const u16* xl[TORGB_MAXJOBS]; };      // blockIdx.y = job: the taps of all blocks against one drgb
__global__ __launch_bounds__(256) void torgb_bwd_w_partial_x3_k512_kernel(TorgbJobs J, const float* __restrict__ drgb,
                                                                          float* __restrict__ partial, long long M) {
  constexpr int K = 512;
  const u16* __restrict__ xh = J.xh[blockIdx.y];
  const u16* __restrict__ xl = J.xl[blockIdx.y];
  __shared__ float sh[3][3][K];                 // waves 1..3
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: drgb comes
  const long long m0 = (long long)blockIdx.x * TORGB_ROWS;                                       // through scalar loads
  const long long m1 = (m0 + TORGB_ROWS < M) ? m0 + TORGB_ROWS : M;
  float acc[3][8];
  float gs[3] = {0.f, 0.f, 0.f};                            // sum of drgb over this wave's rows (the bias gradient)
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[c][q] = 0.f;
  constexpr int RF = 8;                                     // rows in flight per wave: m, m+4, ..., m+28 (4 in round 2:
                                                            // 8 KiB of loads per wave in flight left the kernel at 3 TB/s)
  for (long long m = m0 + wave; m < m1; m += 4 * RF) {
    float v[RF][8], g[RF][3];
#pragma unroll
    for (int r = 0; r < RF; ++r) {
      const long long mm = m + 4 * r;
      const bool ok = mm < m1;
      ld8x3(xh, xl, (ok ? mm : m) * K + lane * 8, v[r]);
#pragma unroll
      for (int c = 0; c < 3; ++c) g[r][c] = ok ? drgb[mm * 3 + c] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < RF; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        gs[c] += g[r][c];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[c][q] = fmaf(g[r][c], v[r][q], acc[c][q]);
      }
  }
  __shared__ float shg[3][3];
  if (wave > 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
      for (int q = 0; q < 8; ++q) sh[wave - 1][c][lane * 8 + q] = acc[c][q];
      if (lane == 0) shg[wave - 1][c] = gs[c];
    }
  }
  __syncthreads();
  float* out = partial + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * 4 * K;
  if (wave == 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float t[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) t[q] = ((acc[c][q] + sh[0][c][lane * 8 + q]) + sh[1][c][lane * 8 + q]) + sh[2][c][lane * 8 + q];
      *reinterpret_cast<float4*>(out + c * K + lane * 8) = make_float4(t[0], t[1], t[2], t[3]);
      *reinterpret_cast<float4*>(out + c * K + lane * 8 + 4) = make_float4(t[4], t[5], t[6], t[7]);
      if (lane == 0) out[3 * K + c] = ((gs[c] + shg[0][c]) + shg[1][c]) + shg[2][c];
    }
  }
}


// partial[chunk][c][k] = sum_{m in chunk} drgb[m][c] x[m][k] ; partial[chunk][3][0..2] = sum drgb
// 256 threads: thread -> 4 consecutive k (float4) x row parity group; K <= 512 per pass.
template <bool X3>
__global__ __launch_bounds__(256) void torgb_bwd_w_partial_kernel(const void* __restrict__ xa, const void* __restrict__ xb,
                                                                  const float* __restrict__ drgb,
                                                                  float* __restrict__ partial, long long M, int K) {
  __shared__ float sh[3][512];
  const long long m0 = (long long)blockIdx.x * TORGB_ROWS;
  const long long m1 = (m0 + TORGB_ROWS < M) ? m0 + TORGB_ROWS : M;
  float* out = partial + (long long)blockIdx.x * 4 * K;
  const int nk4 = K / 4;                       // float4 columns
  const int groups = 256 / (nk4 < 256 ? nk4 : 256) > 0 ? 256 / (nk4 < 256 ? nk4 : 256) : 1;
  for (int kb = 0; kb < nk4; kb += 256) {
    const int cols = (nk4 - kb) < 256 ? (nk4 - kb) : 256;     // float4 columns handled this pass
    const int grp = 256 / cols;                               // row-interleave groups
    const int c4 = threadIdx.x % cols, gi = threadIdx.x / cols;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0;
    if (gi < grp)
      for (long long m = m0 + gi; m < m1; m += grp) {
        const float4 v = ldx4<X3>(xa, xb, m * K + (kb + c4) * 4);
        const float g0 = drgb[m * 3 + 0], g1 = drgb[m * 3 + 1], g2 = drgb[m * 3 + 2];
        a0.x = fmaf(g0, v.x, a0.x); a0.y = fmaf(g0, v.y, a0.y); a0.z = fmaf(g0, v.z, a0.z); a0.w = fmaf(g0, v.w, a0.w);
        a1.x = fmaf(g1, v.x, a1.x); a1.y = fmaf(g1, v.y, a1.y); a1.z = fmaf(g1, v.z, a1.z); a1.w = fmaf(g1, v.w, a1.w);
        a2.x = fmaf(g2, v.x, a2.x); a2.y = fmaf(g2, v.y, a2.y); a2.z = fmaf(g2, v.z, a2.z); a2.w = fmaf(g2, v.w, a2.w);
      }
    // combine row groups deterministically through LDS (group 0 first, then 1, ...)
    for (int g = 0; g < grp; ++g) {
      __syncthreads();
      if (gi == g) {
        float* s0 = &sh[0][c4 * 4]; float* s1 = &sh[1][c4 * 4]; float* s2 = &sh[2][c4 * 4];
        if (g == 0) {
          s0[0] = a0.x; s0[1] = a0.y; s0[2] = a0.z; s0[3] = a0.w;
          s1[0] = a1.x; s1[1] = a1.y; s1[2] = a1.z; s1[3] = a1.w;
          s2[0] = a2.x; s2[1] = a2.y; s2[2] = a2.z; s2[3] = a2.w;
        } else {
          s0[0] += a0.x; s0[1] += a0.y; s0[2] += a0.z; s0[3] += a0.w;
          s1[0] += a1.x; s1[1] += a1.y; s1[2] += a1.z; s1[3] += a1.w;
          s2[0] += a2.x; s2[1] += a2.y; s2[2] += a2.z; s2[3] += a2.w;
        }
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < cols * 4; i += 256) {
      out[kb * 4 + i] = sh[0][i]; out[K + kb * 4 + i] = sh[1][i]; out[2 * K + kb * 4 + i] = sh[2][i];
    }
    __syncthreads();
  }
  (void)groups;
  if (threadIdx.x < 3) {
    float sacc = 0.f;
    for (long long m = m0; m < m1; ++m) sacc += drgb[m * 3 + threadIdx.x];
    out[3 * K + threadIdx.x] = sacc;
  }
}

// sum the per-chunk partials: block -> 8 consecutive outputs x 32 chunk groups (193 blocks at K = 512 instead of 49,
// 32 serial loads per thread instead of 128), fixed combine order
__global__ __launch_bounds__(256) void torgb_bwd_w_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                                 float* __restrict__ dbias, int chunks, int K) {
  __shared__ float red[32][9];
  const int c = threadIdx.x & 7, gi = threadIdx.x >> 3;
  const int idx = blockIdx.x * 8 + c;
  partial += (long long)blockIdx.y * chunks * 4 * K;      // blockIdx.y = job of a batched call
  dw += (long long)blockIdx.y * 3 * K;
  dbias += blockIdx.y * 3;
  float acc = 0.f;
  if (idx < 3 * K + 3)
    for (int ch = gi; ch < chunks; ch += 32) acc += partial[(long long)ch * 4 * K + idx];
  red[gi][c] = acc;
  __syncthreads();
  if (gi == 0 && idx < 3 * K + 3) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 32; ++g) t += red[g][c];
    if (idx < 3 * K) dw[idx] = t; else dbias[idx - 3 * K] = t;
  }
}

// dx[m][k] = sum_c drgb[m][c] w[c][k] (+ add[m][k]); optional unmasked copy; masked by (mask>0 ? 1 : slope)
__global__ __launch_bounds__(256) void torgb_bwd_x_kernel(const float* __restrict__ drgb, const float* __restrict__ w,
                                                          const float* __restrict__ add, const float* __restrict__ mask,
                                                          float slope, float* __restrict__ out_unmasked,
                                                          float* __restrict__ out, long long M, int K) {
  const long long total4 = M * K / 4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
    const long long e = i * 4;
    const long long m = e / K;
    const int k = (int)(e % K);
    const float g0 = drgb[m * 3 + 0], g1 = drgb[m * 3 + 1], g2 = drgb[m * 3 + 2];
    const float4 w0 = *reinterpret_cast<const float4*>(w + k);
    const float4 w1 = *reinterpret_cast<const float4*>(w + K + k);
    const float4 w2 = *reinterpret_cast<const float4*>(w + 2 * K + k);
    float4 v;
    v.x = fmaf(g0, w0.x, fmaf(g1, w1.x, g2 * w2.x));
    v.y = fmaf(g0, w0.y, fmaf(g1, w1.y, g2 * w2.y));
    v.z = fmaf(g0, w0.z, fmaf(g1, w1.z, g2 * w2.z));
    v.w = fmaf(g0, w0.w, fmaf(g1, w1.w, g2 * w2.w));
    if (add) {
      const float4 a = *reinterpret_cast<const float4*>(add + e);
      v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
    }
    if (out_unmasked) *reinterpret_cast<float4*>(out_unmasked + e) = v;
    if (mask) {
      const float4 mk = *reinterpret_cast<const float4*>(mask + e);
      v.x *= mk.x > 0.f ? 1.f : slope; v.y *= mk.y > 0.f ? 1.f : slope;
      v.z *= mk.z > 0.f ? 1.f : slope; v.w *= mk.w > 0.f ? 1.f : slope;
    }
    *reinterpret_cast<float4*>(out + e) = v;
  }
}

// Split-plane form of the same: P (hi, lo planes) = (drgb @ w) * (gate bit ? 1 : slope), gate as a BIT plane (bit k&7 of byte
// [m][k>>3]; NULL: no gate), optional fp32 copy before gating.  One lane = 8 consecutive columns: 3 scalars of drgb, one gate
// byte, two 16-byte stores — a pure 4-bytes-per-element write stream (the round-2 path ran this rank-3 product as a K = 32
// zero-padded bf16x3 GEMM: 138 us + two padding kernels for 268 MB of output at C2; the sums here are exact fp32).
__global__ __launch_bounds__(256) void torgb_bwd_x_x3_kernel(const float* __restrict__ drgb, const float* __restrict__ w,
                                                             const unsigned char* __restrict__ gate, float slope,
                                                             float* __restrict__ out_unmasked, u16* __restrict__ ph,
                                                             u16* __restrict__ pl, long long M, int K) {
  const int k8 = K / 8;
  const long long total = M * k8;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long m = i / k8;
    const int k = (int)(i - m * k8) * 8;
    const float g0 = drgb[m * 3 + 0], g1 = drgb[m * 3 + 1], g2 = drgb[m * 3 + 2];
    float v[8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float4 w0 = *reinterpret_cast<const float4*>(w + k + 4 * h);
      const float4 w1 = *reinterpret_cast<const float4*>(w + K + k + 4 * h);
      const float4 w2 = *reinterpret_cast<const float4*>(w + 2 * K + k + 4 * h);
      v[4 * h + 0] = fmaf(g0, w0.x, fmaf(g1, w1.x, g2 * w2.x));
      v[4 * h + 1] = fmaf(g0, w0.y, fmaf(g1, w1.y, g2 * w2.y));
      v[4 * h + 2] = fmaf(g0, w0.z, fmaf(g1, w1.z, g2 * w2.z));
      v[4 * h + 3] = fmaf(g0, w0.w, fmaf(g1, w1.w, g2 * w2.w));
    }
    const long long e = m * K + k;
    if (out_unmasked) {
      *reinterpret_cast<float4*>(out_unmasked + e) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(out_unmasked + e + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
    if (gate) {
      const unsigned gb = gate[e >> 3];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] *= ((gb >> q) & 1u) ? 1.f : slope;
    }
    unsigned hw[4], lw[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const u16 h0 = f2bf_rne(v[2 * q]), h1 = f2bf_rne(v[2 * q + 1]);
      const u16 l0 = f2bf_rne(v[2 * q] - __uint_as_float(((unsigned)h0) << 16)), l1 = f2bf_rne(v[2 * q + 1] - __uint_as_float(((unsigned)h1) << 16));
      hw[q] = (unsigned)h0 | ((unsigned)h1 << 16);
      lw[q] = (unsigned)l0 | ((unsigned)l1 << 16);
    }
    *reinterpret_cast<uint4*>(ph + e) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
    *reinterpret_cast<uint4*>(pl + e) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
  }
}


// ---- batched forms: every modulated-FC layer of the CIPS head in ONE launch --------------------------------
// The per-layer kernels above are a few microseconds of work each; 18 layers x (2 + 3) launches per step cost more
// in launch tails and idle CUs than in arithmetic.  The weights and styles of all layers are known before the
// head's forward starts, and every dL/dWb is known once its backward is through, so both directions batch.
constexpr int MAXJOBS = 24;
struct PrepJobs {
  const float* W[MAXJOBS]; const float* s[MAXJOBS]; float* demod[MAXJOBS];
  unsigned short *wbh[MAXJOBS], *wbl[MAXJOBS], *wth[MAXJOBS], *wtl[MAXJOBS];
  int in_dim[MAXJOBS], out_dim[MAXJOBS];
};
__global__ __launch_bounds__(256) void modfc_demod_batch_kernel(PrepJobs J, int B, float eps) {
  __shared__ float red[8][33];
  const int job = blockIdx.z, b = blockIdx.y;
  const int in_dim = J.in_dim[job], out_dim = J.out_dim[job];
  if (blockIdx.x * 32 >= out_dim) return;
  const float* __restrict__ W = J.W[job];
  const float* __restrict__ sb = J.s[job] + (long long)b * in_dim;
  const int c = threadIdx.x & 31, kg = threadIdx.x >> 5;
  const int n = blockIdx.x * 32 + c;
  float q = 0.f;
  if (n < out_dim)
    for (int k = kg; k < in_dim; k += 8) {
      float u = W[(long long)k * out_dim + n] * (sb[k] + 1.f);
      q = fmaf(u, u, q);
    }
  red[kg][c] = q;
  __syncthreads();
  if (kg == 0 && n < out_dim) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) t += red[g][c];
    J.demod[job][(long long)b * out_dim + n] = rsqrtf(t + eps);
  }
}
__global__ __launch_bounds__(256) void modfc_planes_batch_kernel(PrepJobs J, int B) {
  __shared__ unsigned short th[32][33], tl[32][33];
  const int job = blockIdx.z / B, b = blockIdx.z % B;
  const int in_dim = J.in_dim[job], out_dim = J.out_dim[job];
  const int n0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
  if (n0 >= out_dim || k0 >= in_dim) return;
  const float* __restrict__ W = J.W[job];
  const float* __restrict__ s = J.s[job];
  const float* __restrict__ demod = J.demod[job];
  unsigned short *wbh = J.wbh[job], *wbl = J.wbl[job], *wth = J.wth[job], *wtl = J.wtl[job];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const long long base = (long long)b * in_dim * out_dim;
  for (int kk = ty; kk < 32; kk += 8) {
    const int k = k0 + kk, n = n0 + tx;
    unsigned short h = 0, l = 0;
    if (k < in_dim && n < out_dim) {
      const float v = W[(long long)k * out_dim + n] * (s[(long long)b * in_dim + k] + 1.f) * demod[(long long)b * out_dim + n];
      h = f2bf_rne(v);
      l = f2bf_rne(v - __uint_as_float(((unsigned)h) << 16));
      wbh[base + (long long)k * out_dim + n] = h;
      wbl[base + (long long)k * out_dim + n] = l;
    }
    th[kk][tx] = h; tl[kk][tx] = l;
  }
  __syncthreads();
  for (int nn = ty; nn < 32; nn += 8) {
    const int n = n0 + nn, k = k0 + tx;
    if (k < in_dim && n < out_dim) {
      wth[base + (long long)n * in_dim + k] = th[tx][nn];
      wtl[base + (long long)n * in_dim + k] = tl[tx][nn];
    }
  }
}

// The same planes in 64 x 64 tiles: float4 reads of W along n, 8-byte plane stores along n straight from registers,
// 16-byte stores of the transposed planes along k through LDS (the 32 x 32 form above writes 64-byte segments of 2-byte
// elements: 3.4 TB/s on the 1.2 GB of per-sample weight planes of a C2 step).  Needs out_dim % 4 == 0, in_dim % 8 == 0.
__global__ __launch_bounds__(256) void modfc_planes_batch64_kernel(PrepJobs J, int B) {
  __shared__ __attribute__((aligned(16))) unsigned short th[64][72], tl[64][72];      // [n local][k local]
  const int job = blockIdx.z / B, b = blockIdx.z % B;
  const int in_dim = J.in_dim[job], out_dim = J.out_dim[job];
  const int n0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
  if (n0 >= out_dim || k0 >= in_dim) return;
  const float* __restrict__ W = J.W[job];
  const float* __restrict__ s = J.s[job];
  const float* __restrict__ demod = J.demod[job];
  unsigned short *wbh = J.wbh[job], *wbl = J.wbl[job], *wth = J.wth[job], *wtl = J.wtl[job];
  const int t = threadIdx.x;
  const long long base = (long long)b * in_dim * out_dim;
  {
    const int col4 = t & 15, r_ = t >> 4;
    const int n = n0 + 4 * col4;
    const bool nok = n < out_dim;
    float4 dm = make_float4(0.f, 0.f, 0.f, 0.f);
    if (nok) dm = *reinterpret_cast<const float4*>(demod + (long long)b * out_dim + n);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int kl = r_ + 16 * p, k = k0 + kl;
      unsigned short h[4] = {0, 0, 0, 0}, l[4] = {0, 0, 0, 0};
      if (nok && k < in_dim) {
        const float4 w = *reinterpret_cast<const float4*>(W + (long long)k * out_dim + n);
        const float sc = s[(long long)b * in_dim + k] + 1.f;
        const float v[4] = {w.x * sc * dm.x, w.y * sc * dm.y, w.z * sc * dm.z, w.w * sc * dm.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          h[j] = f2bf_rne(v[j]);
          l[j] = f2bf_rne(v[j] - __uint_as_float(((unsigned)h[j]) << 16));
        }
        const long long o = base + (long long)k * out_dim + n;
        *reinterpret_cast<uint2*>(wbh + o) = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
        *reinterpret_cast<uint2*>(wbl + o) = make_uint2((unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) { th[4 * col4 + j][kl] = h[j]; tl[4 * col4 + j][kl] = l[j]; }
    }
  }
  __syncthreads();
  {
    const int c8 = t & 7, r_ = t >> 3;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int nl = r_ + 32 * p, n = n0 + nl, k = k0 + 8 * c8;
      if (n < out_dim && k < in_dim) {
        const long long o = base + (long long)n * in_dim + k;
        *reinterpret_cast<uint4*>(wth + o) = *reinterpret_cast<const uint4*>(&th[nl][8 * c8]);
        *reinterpret_cast<uint4*>(wtl + o) = *reinterpret_cast<const uint4*>(&tl[nl][8 * c8]);
      }
    }
  }
}

struct BwdJobs {
  const float* W[MAXJOBS]; const float* s[MAXJOBS]; const float* demod[MAXJOBS]; const float* G[MAXJOBS];
  float *cbuf[MAXJOBS], *dW[MAXJOBS], *ds[MAXJOBS];
  int in_dim[MAXJOBS], out_dim[MAXJOBS];
};
__global__ __launch_bounds__(256) void modfc_prep_bwd_c_batch_kernel(BwdJobs J) {
  __shared__ float red[8][33];
  const int job = blockIdx.z, b = blockIdx.y;
  const int in_dim = J.in_dim[job], out_dim = J.out_dim[job];
  if (blockIdx.x * 32 >= out_dim) return;
  const float* __restrict__ W = J.W[job];
  const float* __restrict__ sb = J.s[job] + (long long)b * in_dim;
  const float* __restrict__ Gb = J.G[job] + (long long)b * in_dim * out_dim;
  const int c = threadIdx.x & 31, kg = threadIdx.x >> 5;
  const int n = blockIdx.x * 32 + c;
  float q = 0.f;
  if (n < out_dim)
    for (int k = kg; k < in_dim; k += 8)
      q = fmaf(Gb[(long long)k * out_dim + n], W[(long long)k * out_dim + n] * (sb[k] + 1.f), q);
  red[kg][c] = q;
  __syncthreads();
  if (kg == 0 && n < out_dim) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) t += red[g][c];
    J.cbuf[job][(long long)b * out_dim + n] = t;
  }
}
__global__ __launch_bounds__(256) void modfc_prep_bwd_w_batch_kernel(BwdJobs J, int B) {
  const int job = blockIdx.y;
  const int in_dim = J.in_dim[job], out_dim = J.out_dim[job];
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)in_dim * out_dim) return;
  const float* __restrict__ s = J.s[job]; const float* __restrict__ demod = J.demod[job];
  const float* __restrict__ G = J.G[job]; const float* __restrict__ cbuf = J.cbuf[job];
  const int k = (int)(idx / out_dim), n = (int)(idx % out_dim);
  const float w = J.W[job][idx];
  float acc = 0.f;
  for (int b = 0; b < B; ++b) {   // fixed order: deterministic
    const float m = s[(long long)b * in_dim + k] + 1.f;
    const float d = demod[(long long)b * out_dim + n];
    const float g = G[((long long)b * in_dim + k) * out_dim + n];
    const float du = d * (g - d * d * (w * m) * cbuf[(long long)b * out_dim + n]);
    acc = fmaf(m, du, acc);
  }
  J.dW[job][idx] = acc;
}
__global__ __launch_bounds__(256) void modfc_prep_bwd_s_batch_kernel(BwdJobs J, int B) {
  const int job = blockIdx.y;
  const int in_dim = J.in_dim[job], out_dim = J.out_dim[job];
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long long)B * in_dim) return;
  const float* __restrict__ W = J.W[job]; const float* __restrict__ demod = J.demod[job];
  const float* __restrict__ G = J.G[job]; const float* __restrict__ cbuf = J.cbuf[job];
  const int b = (int)(row / in_dim), k = (int)(row % in_dim);
  const float m = J.s[job][row] + 1.f;
  float acc = 0.f;
  for (int n = lane; n < out_dim; n += 64) {
    const float w = W[(long long)k * out_dim + n];
    const float d = demod[(long long)b * out_dim + n];
    const float g = G[row * out_dim + n];
    const float du = d * (g - d * d * (w * m) * cbuf[(long long)b * out_dim + n]);
    acc = fmaf(w, du, acc);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
  if (lane == 0) J.ds[job][row] = acc;
}

// ---- vector forms of the column reductions and the fused dW / ds pass (round 3) ------------------------------------
// The per-image gradient G = dL/dWb of the 18 head layers is 604 MB at C2.  The three-pass backward above reads it
// three times (c, dW, ds); here the column sums c read it once with 16-byte lanes and ONE further pass produces both
// dW (sum over images, registers) and ds (sum over columns, one wave reduction per image) — two reads instead of three.
// A block is 64 columns x 16 row groups, every lane owns 4 consecutive columns.
struct ColJobs {                       // one table for both uses: G == nullptr selects the demodulation sums
  const float* W[MAXJOBS]; const float* s[MAXJOBS]; const float* G[MAXJOBS]; float* out[MAXJOBS];
  int in_dim[MAXJOBS], out_dim[MAXJOBS];
};
template <bool WITH_G>
__global__ __launch_bounds__(256) void modfc_colsum4_batch_kernel(ColJobs J, int B, float eps) {
  __shared__ float4 red[16][17];
  const int job = blockIdx.z, b = blockIdx.y;
  const int in_dim = J.in_dim[job], out_dim = J.out_dim[job];
  if (blockIdx.x * 64 >= out_dim) return;
  const float* __restrict__ W = J.W[job];
  const float* __restrict__ sb = J.s[job] + (long long)b * in_dim;
  const float* __restrict__ Gb = WITH_G ? J.G[job] + (long long)b * in_dim * out_dim : nullptr;
  const int c = threadIdx.x & 15, kg = threadIdx.x >> 4;
  const int n = blockIdx.x * 64 + 4 * c;
  float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
  if (n < out_dim) {
#pragma unroll 4
    for (int k = kg; k < in_dim; k += 16) {
      const float4 w = *reinterpret_cast<const float4*>(W + (long long)k * out_dim + n);
      const float m = sb[k] + 1.f;
      if (WITH_G) {
        const float4 g = *reinterpret_cast<const float4*>(Gb + (long long)k * out_dim + n);
        q.x = fmaf(g.x, w.x * m, q.x); q.y = fmaf(g.y, w.y * m, q.y); q.z = fmaf(g.z, w.z * m, q.z); q.w = fmaf(g.w, w.w * m, q.w);
      } else {
        const float4 u = make_float4(w.x * m, w.y * m, w.z * m, w.w * m);
        q.x = fmaf(u.x, u.x, q.x); q.y = fmaf(u.y, u.y, q.y); q.z = fmaf(u.z, u.z, q.z); q.w = fmaf(u.w, u.w, q.w);
      }
    }
  }
  red[kg][c] = q;
  __syncthreads();
  if (kg == 0 && n < out_dim) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int g = 0; g < 16; ++g) { const float4 v = red[g][c]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
    if (WITH_G) *reinterpret_cast<float4*>(J.out[job] + (long long)b * out_dim + n) = t;
    else *reinterpret_cast<float4*>(J.out[job] + (long long)b * out_dim + n) =
        make_float4(rsqrtf(t.x + eps), rsqrtf(t.y + eps), rsqrtf(t.z + eps), rsqrtf(t.w + eps));
  }
}

// one wave per weight row k, lanes own 4 consecutive columns of each 256-column chunk (out_dim <= 256 NCH, B <= 64):
// dW[k][n] = sum_b m_bk du_bkn in registers (images in ascending order, as the three-pass kernel adds them), and
// ds[b][k] = sum_n W_kn du_bkn as one wave reduction per image, kept by lane b.
template <int NCH>
__global__ __launch_bounds__(256) void modfc_prep_bwd_ws_batch_kernel(BwdJobs J, int B) {
  const int job = blockIdx.y;
  const int in_dim = J.in_dim[job], out_dim = J.out_dim[job];
  const int lane = threadIdx.x & 63;
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= in_dim) return;
  const float* __restrict__ s = J.s[job]; const float* __restrict__ demod = J.demod[job];
  const float* __restrict__ G = J.G[job]; const float* __restrict__ cbuf = J.cbuf[job];
  float4 w[NCH], acc[NCH];
  bool ok[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int n = c * 256 + 4 * lane;
    ok[c] = n < out_dim;
    w[c] = ok[c] ? *reinterpret_cast<const float4*>(J.W[job] + (long long)k * out_dim + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float dsv = 0.f;
  for (int b = 0; b < B; ++b) {
    const float m = s[(long long)b * in_dim + k] + 1.f;
    const long long row = (long long)b * in_dim + k;
    float part = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if (!ok[c]) continue;
      const int n = c * 256 + 4 * lane;
      const float4 g = *reinterpret_cast<const float4*>(G + row * out_dim + n);
      const float4 d = *reinterpret_cast<const float4*>(demod + (long long)b * out_dim + n);
      const float4 cb = *reinterpret_cast<const float4*>(cbuf + (long long)b * out_dim + n);
      const float dx = d.x * (g.x - d.x * d.x * (w[c].x * m) * cb.x), dy = d.y * (g.y - d.y * d.y * (w[c].y * m) * cb.y);
      const float dz = d.z * (g.z - d.z * d.z * (w[c].z * m) * cb.z), dw = d.w * (g.w - d.w * d.w * (w[c].w * m) * cb.w);
      acc[c].x = fmaf(m, dx, acc[c].x); acc[c].y = fmaf(m, dy, acc[c].y); acc[c].z = fmaf(m, dz, acc[c].z); acc[c].w = fmaf(m, dw, acc[c].w);
      part = fmaf(w[c].x, dx, part); part = fmaf(w[c].y, dy, part); part = fmaf(w[c].z, dz, part); part = fmaf(w[c].w, dw, part);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off);
    if (lane == b) dsv = part;
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c)
    if (ok[c]) *reinterpret_cast<float4*>(J.dW[job] + (long long)k * out_dim + c * 256 + 4 * lane) = acc[c];
  if (lane < B) J.ds[job][(long long)lane * in_dim + k] = dsv;
}


// ---- co-resident forms (round 6) ----------------------------------------------------------------------------------
// The fused SIREN backward holds every CU for 2.4 ms with the whole LDS (160 KiB) and 472 of the 512 registers of each SIMD
// lane: a kernel that uses NO LDS and at most 40 VGPRs still gets a wave slot beside it (profiles/r6_coresidency_probe.txt:
// LDS-free launches on a side stream complete at their usual latency while it runs, LDS users wait for it to end).  The
// style / ToRGB weight-gradient tail of the INR head's backward depends on nothing the SIREN backward produces, so these
// forms of its three streaming kernels let it run underneath: one wave per work item, reductions by shuffles and by
// fixed-order partial sums in HBM, few loads in flight per wave (the waves are many).
// loads addressed as SGPR base + 32-bit VGPR byte offset, issued by hand: hipcc otherwise carries one 64-bit VGPR address per load
// stream through these loops and the kernels leave the 40-register budget.  The caller waits with cores_wait().
typedef float f4v __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f4v cores_ld16(const void* base, unsigned voff) {
  f4v r;
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r) : "v"(voff), "s"(base) : "memory");
  return r;
}
__device__ __forceinline__ u2v cores_ld8(const void* base, unsigned voff) {
  u2v r;
  asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(r) : "v"(voff), "s"(base) : "memory");
  return r;
}

constexpr int CS_KS = 4;                 // row segments of the column sums: cbuf holds CS_KS partial planes (CS_KS, B, out)

__global__ __launch_bounds__(64) void modfc_colsum4_cores_kernel(ColJobs J, int B) {
  const int job = blockIdx.z / CS_KS, seg = blockIdx.z % CS_KS, b = blockIdx.y;
  const int in_dim = J.in_dim[job], out_dim = J.out_dim[job];
  const int n = blockIdx.x * 256 + 4 * threadIdx.x;
  if (n >= out_dim) return;
  const int k0 = (int)((long long)in_dim * seg / CS_KS), k1 = (int)((long long)in_dim * (seg + 1) / CS_KS);
  const float* __restrict__ W = J.W[job] + n;
  const float* __restrict__ sb = J.s[job] + (long long)b * in_dim;
  const float* __restrict__ Gb = J.G[job] + (long long)b * in_dim * out_dim + n;
  float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 2
  for (int k = k0; k < k1; ++k) {
    const float4 w = *reinterpret_cast<const float4*>(W + (long long)k * out_dim);
    const float4 g = *reinterpret_cast<const float4*>(Gb + (long long)k * out_dim);
    const float m = sb[k] + 1.f;
    q.x = fmaf(g.x, w.x * m, q.x); q.y = fmaf(g.y, w.y * m, q.y); q.z = fmaf(g.z, w.z * m, q.z); q.w = fmaf(g.w, w.w * m, q.w);
  }
  *reinterpret_cast<float4*>(J.out[job] + ((long long)seg * B + b) * out_dim + n) = q;
}

// cbuf plane 0 += planes 1 .. CS_KS-1 (fixed order): the column sums the next kernel reads
__global__ __launch_bounds__(256) void modfc_colsum_fold_cores_kernel(ColJobs J, int B) {
  const int job = blockIdx.y;
  const int n4 = B * J.out_dim[job] / 4;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4* c = reinterpret_cast<float4*>(J.out[job]);
  float4 a = c[i];
#pragma unroll
  for (int j = 1; j < CS_KS; ++j) { const float4 t = c[j * n4 + i]; a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w; }
  c[i] = a;
}

// one wave per weight row k, one 256-column chunk at a time (registers are reused from chunk to chunk; lane b keeps ds[b][k]);
// uniform base pointers + one per-lane offset, so the loads address as SGPR base + VGPR offset
__global__ __launch_bounds__(256) void modfc_prep_bwd_ws_cores_kernel(BwdJobs J, int B) {
  const int job = blockIdx.y;
  const int in_dim = J.in_dim[job], out_dim = J.out_dim[job];
  const int lane = threadIdx.x & 63;
  const int k = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (k >= in_dim) return;
  const float* __restrict__ s = J.s[job] + k;
  const int gstep = in_dim * out_dim;
  float dsv = 0.f;
#pragma unroll 1
  for (int c0 = 0; c0 < out_dim; c0 += 256) {
    const bool ok = c0 + 4 * lane < out_dim;
    const float4* __restrict__ Wp = reinterpret_cast<const float4*>(J.W[job] + k * out_dim + c0);
    const float4* __restrict__ Gp = reinterpret_cast<const float4*>(J.G[job] + k * out_dim + c0);
    const float4* __restrict__ Dp = reinterpret_cast<const float4*>(J.demod[job] + c0);
    const float4* __restrict__ Cp = reinterpret_cast<const float4*>(J.cbuf[job] + c0);
    const float4 w = ok ? Wp[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const unsigned voff = ok ? 16u * lane : 0u;           // lanes past the row's end read (and discard) its first columns
#pragma unroll 1
    for (int b = 0; b < B; ++b) {
      const float m = s[b * in_dim] + 1.f;
      f4v g = cores_ld16(Gp, voff), d = cores_ld16(Dp, voff), cb = cores_ld16(Cp, voff);
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(g), "+v"(d), "+v"(cb) :: "memory");
      const float dx = d.x * (g.x - d.x * d.x * (w.x * m) * cb.x), dy = d.y * (g.y - d.y * d.y * (w.y * m) * cb.y);
      const float dz = d.z * (g.z - d.z * d.z * (w.z * m) * cb.z), dw = d.w * (g.w - d.w * d.w * (w.w * m) * cb.w);
      float part = 0.f;
      if (ok) {
        acc.x = fmaf(m, dx, acc.x); acc.y = fmaf(m, dy, acc.y); acc.z = fmaf(m, dz, acc.z); acc.w = fmaf(m, dw, acc.w);
        part = fmaf(w.x, dx, part); part = fmaf(w.y, dy, part); part = fmaf(w.z, dz, part); part = fmaf(w.w, dw, part);
      }
      Gp += gstep / 4; Dp += out_dim / 4; Cp += out_dim / 4;
      const float tot = wave_sum_dpp(part);
      if (lane == b) dsv += tot;
    }
    if (ok) reinterpret_cast<float4*>(J.dW[job] + k * out_dim + c0)[lane] = acc;
  }
  if (lane < B) J.ds[job][lane * in_dim + k] = dsv;
}

// ToRGB weight gradient, K = 512: a wave owns one 128-row chunk and one half of the 512 columns (4 per lane), rows in order,
// two in flight; drgb through scalar loads.  Same partial layout as torgb_bwd_w_partial_x3_k512_kernel.
__global__ __launch_bounds__(256) void torgb_bwd_w_partial_cores_kernel(TorgbJobs J, const float* __restrict__ drgb,
                                                                        float* __restrict__ partial, long long M, int chunks) {
  constexpr int K = 512;
  const u16* __restrict__ xh = J.xh[blockIdx.y];
  const u16* __restrict__ xl = J.xl[blockIdx.y];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int chunk = blockIdx.x * 2 + (wave >> 1);
  if (chunk >= chunks) return;
  const int kk = (wave & 1) * 256 + lane * 4;
  const long long m0 = (long long)chunk * TORGB_ROWS;
  const long long m1 = (m0 + TORGB_ROWS < M) ? m0 + TORGB_ROWS : M;
  float acc[3][4];
  float gs[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[c][q] = 0.f;
#pragma unroll 1
  for (long long m = m0; m < m1; m += 2) {
    const bool two = m + 1 < m1;
    const long long mb = two ? m + 1 : m;
    u2v ha = cores_ld8(xh + m * K, 2u * kk), la = cores_ld8(xl + m * K, 2u * kk);
    u2v hb = cores_ld8(xh + mb * K, 2u * kk), lb = cores_ld8(xl + mb * K, 2u * kk);
    float ga[3], gb[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { ga[c] = drgb[m * 3 + c]; gb[c] = two ? drgb[mb * 3 + c] : 0.f; }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(ha), "+v"(la), "+v"(hb), "+v"(lb) :: "memory");
    const float va[4] = {__uint_as_float(ha.x << 16) + __uint_as_float(la.x << 16), __uint_as_float(ha.x & 0xffff0000u) + __uint_as_float(la.x & 0xffff0000u),
                         __uint_as_float(ha.y << 16) + __uint_as_float(la.y << 16), __uint_as_float(ha.y & 0xffff0000u) + __uint_as_float(la.y & 0xffff0000u)};
    const float vb[4] = {__uint_as_float(hb.x << 16) + __uint_as_float(lb.x << 16), __uint_as_float(hb.x & 0xffff0000u) + __uint_as_float(lb.x & 0xffff0000u),
                         __uint_as_float(hb.y << 16) + __uint_as_float(lb.y << 16), __uint_as_float(hb.y & 0xffff0000u) + __uint_as_float(lb.y & 0xffff0000u)};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      gs[c] += ga[c]; gs[c] += gb[c];
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[c][q] = fmaf(gb[c], vb[q], fmaf(ga[c], va[q], acc[c][q]));
    }
  }
  float* out = partial + ((long long)blockIdx.y * chunks + chunk) * 4 * K;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    *reinterpret_cast<float4*>(out + c * K + kk) = make_float4(acc[c][0], acc[c][1], acc[c][2], acc[c][3]);
    if (kk == 0) out[3 * K + c] = gs[c];
  }
}

// partial (njobs, chunks, 4, K) -> dw (njobs, 3, K), dbias (njobs, 3): a wave owns 4 consecutive entries, its lanes take the
// chunks lane, lane + 64, ... in order and are combined by a shuffle tree (fixed order)
__global__ __launch_bounds__(64) void torgb_bwd_w_reduce_cores_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                                      float* __restrict__ dbias, int chunks, int K) {
  const int idx = blockIdx.x * 4;                         // 0 .. 3K (+3 bias sums): 3K is a multiple of 4
  partial += (long long)blockIdx.y * chunks * 4 * K;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int ch = threadIdx.x; ch < chunks; ch += 64) {
    const float4 t = *reinterpret_cast<const float4*>(partial + (long long)ch * 4 * K + idx);
    a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    a.x += __shfl_xor(a.x, off); a.y += __shfl_xor(a.y, off); a.z += __shfl_xor(a.z, off); a.w += __shfl_xor(a.w, off);
  }
  if (threadIdx.x == 0) {
    if (idx < 3 * K) *reinterpret_cast<float4*>(dw + (long long)blockIdx.y * 3 * K + idx) = a;
    else { float* o = dbias + blockIdx.y * 3; o[0] = a.x; o[1] = a.y; o[2] = a.z; }
  }
}

}  // namespace

extern "C" int cips_modfc_prep(const float* weight, const float* s, float* wb, float* wbt, float* demod,
                               int B, int in_dim, int out_dim, float eps, cips_stream_t stream) {
  if (B <= 0 || in_dim <= 0 || out_dim <= 0) return (int)hipErrorInvalidValue;
  dim3 grid((out_dim + 31) / 32, B);
  hipLaunchKernelGGL(modfc_prep_kernel, grid, dim3(256), 0, (hipStream_t)stream, weight, s, wb, wbt, demod,
                     in_dim, out_dim, eps);
  return CIPS_CHECK_LAUNCH();
}

extern "C" int cips_modfc_prep_bwd(const float* weight, const float* s, const float* demod, const float* gwb,
                                   float* cbuf, float* dweight, float* ds, int B, int in_dim, int out_dim,
                                   cips_stream_t stream) {
  if (B <= 0 || in_dim <= 0 || out_dim <= 0) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(modfc_prep_bwd_c_kernel, dim3((out_dim + 31) / 32, B), dim3(256), 0, st, weight, s, gwb,
                     cbuf, in_dim, out_dim);
  long long nw = (long long)in_dim * out_dim;
  hipLaunchKernelGGL(modfc_prep_bwd_w_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, st, weight, s,
                     demod, gwb, cbuf, dweight, B, in_dim, out_dim);
  long long rows = (long long)B * in_dim;
  hipLaunchKernelGGL(modfc_prep_bwd_s_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, weight, s,
                     demod, gwb, cbuf, ds, B, in_dim, out_dim);
  return CIPS_CHECK_LAUNCH();
}

extern "C" int cips_torgb_fwd(const float* x, const float* w, const float* bias, float* rgb, long long M,
                              int K, int accumulate, cips_stream_t stream) {
  if (M <= 0 || K <= 0 || (K & 3)) return (int)hipErrorInvalidValue;
  long long blocks = (M + 3) / 4;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(torgb_fwd_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     (const void*)x, (const void*)nullptr, w, bias, rgb, M, K, accumulate);
  return CIPS_CHECK_LAUNCH();
}

// ToRGB forward folded into the GEMM epilogue (gemm_bf16x3_v3.hip, RGBF): the column-block partials of every row are
// added here in block order, with the bias and the running image
__global__ __launch_bounds__(256) void torgb_finish_kernel(const float4* __restrict__ part, int nblocks, const float* __restrict__ bias,
                                                           float* __restrict__ rgb, long long M, int accumulate) {
  const float b0 = bias ? bias[0] : 0.f, b1 = bias ? bias[1] : 0.f, b2 = bias ? bias[2] : 0.f;
  for (long long m = (long long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long long)gridDim.x * 256) {
    float4 t = part[m];
    for (int j = 1; j < nblocks; ++j) { const float4 u = part[(long long)j * M + m]; t.x += u.x; t.y += u.y; t.z += u.z; }
    float* o = rgb + m * 3;
    const float r0 = t.x + b0, r1 = t.y + b1, r2 = t.z + b2;
    if (accumulate) { o[0] += r0; o[1] += r1; o[2] += r2; } else { o[0] = r0; o[1] = r1; o[2] = r2; }
  }
}
extern "C" int cips_torgb_finish(const float* part, int nblocks, const float* bias, float* rgb, long long M, int accumulate,
                                 cips_stream_t stream) {
  if (!part || !rgb || nblocks <= 0 || M <= 0 || ((uintptr_t)part & 15)) return (int)hipErrorInvalidValue;
  long long blocks = (M + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(torgb_finish_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(part), nblocks, bias, rgb, M, accumulate);
  return CIPS_CHECK_LAUNCH();
}

extern "C" int cips_torgb_fwd_x3(const void* x_hi, const void* x_lo, const float* w, const float* bias, float* rgb,
                                 long long M, int K, int accumulate, cips_stream_t stream) {
  if (M <= 0 || K <= 0 || (K & 3)) return (int)hipErrorInvalidValue;
  long long blocks = (M + 3) / 4;
  if (blocks > 16384) blocks = 16384;
  if (K == 512) {
    long long b4 = (M + 15) / 16;                      // 4 waves x 4 rows per iteration
    if (b4 > 8192) b4 = 8192;
    hipLaunchKernelGGL(torgb_fwd_x3_k512_kernel, dim3((unsigned)b4), dim3(256), 0, (hipStream_t)stream, (const u16*)x_hi,
                       (const u16*)x_lo, w, bias, rgb, M, accumulate);
    return CIPS_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(torgb_fwd_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x_hi, x_lo,
                     w, bias, rgb, M, K, accumulate);
  return CIPS_CHECK_LAUNCH();
}

extern "C" int cips_modfc_prep_x3(const float* weight, const float* s, void* wb_hi, void* wb_lo, void* wbt_hi,
                                  void* wbt_lo, float* demod, int B, int in_dim, int out_dim, float eps,
                                  cips_stream_t stream) {
  if (B <= 0 || in_dim <= 0 || out_dim <= 0) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(modfc_demod_kernel, dim3((out_dim + 31) / 32, B), dim3(256), 0, st, weight, s, demod, in_dim,
                     out_dim, eps);
  hipLaunchKernelGGL(modfc_planes_kernel, dim3((out_dim + 31) / 32, (in_dim + 31) / 32, B), dim3(256), 0, st, weight,
                     s, demod, (unsigned short*)wb_hi, (unsigned short*)wb_lo, (unsigned short*)wbt_hi,
                     (unsigned short*)wbt_lo, in_dim, out_dim);
  return CIPS_CHECK_LAUNCH();
}

extern "C" int cips_torgb_bwd_partials(long long M) { return (int)((M + TORGB_ROWS - 1) / TORGB_ROWS); }

extern "C" int cips_torgb_bwd_w(const float* x, const float* drgb, float* partials, float* dw, float* dbias,
                                long long M, int K, cips_stream_t stream) {
  if (M <= 0 || K <= 0) return (int)hipErrorInvalidValue;
  int chunks = cips_torgb_bwd_partials(M);
  hipStream_t st = (hipStream_t)stream;
  if ((K & 3) || K > 512) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(torgb_bwd_w_partial_kernel<false>, dim3(chunks), dim3(256), 0, st, (const void*)x,
                     (const void*)nullptr, drgb, partials, M, K);
  hipLaunchKernelGGL(torgb_bwd_w_reduce_kernel, dim3((3 * K + 3 + 7) / 8), dim3(256), 0, st, partials, dw,
                     dbias, chunks, K);
  return CIPS_CHECK_LAUNCH();
}

extern "C" int cips_torgb_bwd_w_x3(const void* x_hi, const void* x_lo, const float* drgb, float* partials,
                                   float* dw, float* dbias, long long M, int K, cips_stream_t stream) {
  if (M <= 0 || K <= 0 || (K & 3) || K > 512) return (int)hipErrorInvalidValue;
  int chunks = cips_torgb_bwd_partials(M);
  hipStream_t st = (hipStream_t)stream;
  if (K == 512) {
    TorgbJobs J = {};
    J.xh[0] = (const u16*)x_hi; J.xl[0] = (const u16*)x_lo;
    hipLaunchKernelGGL(torgb_bwd_w_partial_x3_k512_kernel, dim3(chunks), dim3(256), 0, st, J, drgb, partials, M);
  } else
    hipLaunchKernelGGL(torgb_bwd_w_partial_kernel<true>, dim3(chunks), dim3(256), 0, st, x_hi, x_lo, drgb, partials,
                       M, K);
  hipLaunchKernelGGL(torgb_bwd_w_reduce_kernel, dim3((3 * K + 3 + 7) / 8), dim3(256), 0, st, partials, dw,
                     dbias, chunks, K);
  return CIPS_CHECK_LAUNCH();
}

// the ToRGB taps of several blocks (same M, K = 512) against one drgb in two launches: partials (njobs, chunks, 4, K),
// dw (njobs, 3, K), dbias (njobs, 3)
extern "C" int cips_torgb_bwd_w_x3_batch(const void* const* x_hi, const void* const* x_lo, int njobs, const float* drgb,
                                         float* partials, float* dw, float* dbias, long long M, int K, cips_stream_t stream) {
  if (!x_hi || !x_lo || njobs <= 0 || M <= 0) return (int)hipErrorInvalidValue;
  if (njobs > TORGB_MAXJOBS || K != 512) return (int)hipErrorNotSupported;
  TorgbJobs J = {};
  for (int i = 0; i < njobs; ++i) {
    if (!x_hi[i] || !x_lo[i]) return (int)hipErrorInvalidValue;
    J.xh[i] = (const u16*)x_hi[i]; J.xl[i] = (const u16*)x_lo[i];
  }
  const int chunks = cips_torgb_bwd_partials(M);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(torgb_bwd_w_partial_x3_k512_kernel, dim3(chunks, njobs), dim3(256), 0, st, J, drgb, partials, M);
  hipLaunchKernelGGL(torgb_bwd_w_reduce_kernel, dim3((3 * K + 3 + 7) / 8, njobs), dim3(256), 0, st, partials, dw, dbias, chunks, K);
  return CIPS_CHECK_LAUNCH();
}

extern "C" int cips_cores_colsum_parts(void) { return CS_KS; }

// co-resident form of cips_modfc_prep_bwd_batch (no LDS, <= 40 VGPRs: see the kernels): cbuf is (cips_cores_colsum_parts(), B, out)
extern "C" int cips_modfc_prep_bwd_batch_cores(const cips_modfc_bwd_job* jobs, int njobs, int B, cips_stream_t stream) {
  if (!jobs || njobs <= 0 || njobs > MAXJOBS || B <= 0) return (int)hipErrorInvalidValue;
  if (B > 64) return (int)hipErrorNotSupported;
  BwdJobs J;
  ColJobs Cj = {};
  int max_in = 0, max_out = 0;
  for (int i = 0; i < njobs; ++i) {
    const cips_modfc_bwd_job& j = jobs[i];
    if (j.in_dim <= 0 || j.out_dim <= 0) return (int)hipErrorInvalidValue;
    if (j.out_dim % 4) return (int)hipErrorNotSupported;
    J.W[i] = j.weight; J.s[i] = j.s; J.demod[i] = j.demod; J.G[i] = j.gwb;
    J.cbuf[i] = j.cbuf; J.dW[i] = j.dweight; J.ds[i] = j.ds;
    J.in_dim[i] = j.in_dim; J.out_dim[i] = j.out_dim;
    Cj.W[i] = j.weight; Cj.s[i] = j.s; Cj.G[i] = j.gwb; Cj.out[i] = j.cbuf; Cj.in_dim[i] = j.in_dim; Cj.out_dim[i] = j.out_dim;
    max_in = j.in_dim > max_in ? j.in_dim : max_in; max_out = j.out_dim > max_out ? j.out_dim : max_out;
  }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(modfc_colsum4_cores_kernel, dim3((max_out + 255) / 256, B, njobs * CS_KS), dim3(64), 0, st, Cj, B);
  hipLaunchKernelGGL(modfc_colsum_fold_cores_kernel, dim3((B * max_out / 4 + 255) / 256, njobs), dim3(256), 0, st, Cj, B);
  hipLaunchKernelGGL(modfc_prep_bwd_ws_cores_kernel, dim3((max_in + 3) / 4, njobs), dim3(256), 0, st, J, B);
  return CIPS_CHECK_LAUNCH();
}

// co-resident form of cips_torgb_bwd_w_x3_batch: same arguments, layouts and partial scratch
extern "C" int cips_torgb_bwd_w_x3_batch_cores(const void* const* x_hi, const void* const* x_lo, int njobs, const float* drgb,
                                               float* partials, float* dw, float* dbias, long long M, int K, cips_stream_t stream) {
  if (!x_hi || !x_lo || njobs <= 0 || M <= 0) return (int)hipErrorInvalidValue;
  if (njobs > TORGB_MAXJOBS || K != 512) return (int)hipErrorNotSupported;
  TorgbJobs J = {};
  for (int i = 0; i < njobs; ++i) {
    if (!x_hi[i] || !x_lo[i]) return (int)hipErrorInvalidValue;
    J.xh[i] = (const u16*)x_hi[i]; J.xl[i] = (const u16*)x_lo[i];
  }
  const int chunks = cips_torgb_bwd_partials(M);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(torgb_bwd_w_partial_cores_kernel, dim3((chunks + 1) / 2, njobs), dim3(256), 0, st, J, drgb, partials, M, chunks);
  hipLaunchKernelGGL(torgb_bwd_w_reduce_cores_kernel, dim3((3 * K + 4) / 4, njobs), dim3(64), 0, st, partials, dw, dbias, chunks, K);
  return CIPS_CHECK_LAUNCH();
}


extern "C" int cips_torgb_bwd_x(const float* drgb, const float* w, const float* add, const float* mask,
                                float slope, float* out_unmasked, float* out, long long M, int K,
                                cips_stream_t stream) {
  if (M <= 0 || K <= 0 || (K & 3)) return (int)hipErrorInvalidValue;
  long long blocks = (M * K / 4 + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(torgb_bwd_x_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, drgb, w, add,
                     mask, slope, out_unmasked, out, M, K);
  return CIPS_CHECK_LAUNCH();
}

extern "C" int cips_torgb_bwd_x_x3(const float* drgb, const float* w, const void* gate_bits, float slope, float* out_unmasked,
                                   void* p_hi, void* p_lo, long long M, int K, cips_stream_t stream) {
  if (!drgb || !w || !p_hi || !p_lo || M <= 0 || K <= 0 || (K & 7)) return (int)hipErrorInvalidValue;
  long long blocks = (M * (K / 8) + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(torgb_bwd_x_x3_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, drgb, w,
                     (const unsigned char*)gate_bits, slope, out_unmasked, (u16*)p_hi, (u16*)p_lo, M, K);
  return CIPS_CHECK_LAUNCH();
}

extern "C" int cips_modfc_max_jobs(void) { return MAXJOBS; }

extern "C" int cips_modfc_prep_x3_batch(const cips_modfc_prep_job* jobs, int njobs, int B, float eps, cips_stream_t stream) {
  if (!jobs || njobs <= 0 || njobs > MAXJOBS || B <= 0) return (int)hipErrorInvalidValue;
  PrepJobs J;
  int max_in = 0, max_out = 0;
  for (int i = 0; i < njobs; ++i) {
    const cips_modfc_prep_job& j = jobs[i];
    if (j.in_dim <= 0 || j.out_dim <= 0) return (int)hipErrorInvalidValue;
    J.W[i] = j.weight; J.s[i] = j.s; J.demod[i] = j.demod;
    J.wbh[i] = (unsigned short*)j.wb_hi; J.wbl[i] = (unsigned short*)j.wb_lo;
    J.wth[i] = (unsigned short*)j.wbt_hi; J.wtl[i] = (unsigned short*)j.wbt_lo;
    J.in_dim[i] = j.in_dim; J.out_dim[i] = j.out_dim;
    max_in = j.in_dim > max_in ? j.in_dim : max_in; max_out = j.out_dim > max_out ? j.out_dim : max_out;
  }
  hipStream_t st = (hipStream_t)stream;
  bool wide = true;                              // 64 x 64 tiles with vector accesses when every job's shape allows
  bool vec4 = true;                              // 16-byte lanes in the column reductions
  for (int i = 0; i < njobs; ++i) {
    wide = wide && (jobs[i].out_dim % 4 == 0) && (jobs[i].in_dim % 8 == 0);
    vec4 = vec4 && (jobs[i].out_dim % 4 == 0);
  }
  if (vec4) {
    ColJobs Cj = {};
    for (int i = 0; i < njobs; ++i) { Cj.W[i] = J.W[i]; Cj.s[i] = J.s[i]; Cj.out[i] = J.demod[i]; Cj.in_dim[i] = J.in_dim[i]; Cj.out_dim[i] = J.out_dim[i]; }
    hipLaunchKernelGGL(modfc_colsum4_batch_kernel<false>, dim3((max_out + 63) / 64, B, njobs), dim3(256), 0, st, Cj, B, eps);
  } else
    hipLaunchKernelGGL(modfc_demod_batch_kernel, dim3((max_out + 31) / 32, B, njobs), dim3(256), 0, st, J, B, eps);
  if (wide)
    hipLaunchKernelGGL(modfc_planes_batch64_kernel, dim3((max_out + 63) / 64, (max_in + 63) / 64, B * njobs), dim3(256), 0, st, J, B);
  else
    hipLaunchKernelGGL(modfc_planes_batch_kernel, dim3((max_out + 31) / 32, (max_in + 31) / 32, B * njobs), dim3(256), 0, st, J, B);
  return CIPS_CHECK_LAUNCH();
}

extern "C" int cips_modfc_prep_bwd_batch(const cips_modfc_bwd_job* jobs, int njobs, int B, cips_stream_t stream) {
  if (!jobs || njobs <= 0 || njobs > MAXJOBS || B <= 0) return (int)hipErrorInvalidValue;
  BwdJobs J;
  int max_in = 0, max_out = 0;
  long long max_nw = 0;
  for (int i = 0; i < njobs; ++i) {
    const cips_modfc_bwd_job& j = jobs[i];
    if (j.in_dim <= 0 || j.out_dim <= 0) return (int)hipErrorInvalidValue;
    J.W[i] = j.weight; J.s[i] = j.s; J.demod[i] = j.demod; J.G[i] = j.gwb;
    J.cbuf[i] = j.cbuf; J.dW[i] = j.dweight; J.ds[i] = j.ds;
    J.in_dim[i] = j.in_dim; J.out_dim[i] = j.out_dim;
    max_in = j.in_dim > max_in ? j.in_dim : max_in; max_out = j.out_dim > max_out ? j.out_dim : max_out;
    const long long nw = (long long)j.in_dim * j.out_dim;
    max_nw = nw > max_nw ? nw : max_nw;
  }
  hipStream_t st = (hipStream_t)stream;
  bool vec4 = B <= 64 && max_out <= 1024;        // two reads of G instead of three (see modfc_prep_bwd_ws_batch_kernel)
  for (int i = 0; i < njobs; ++i) vec4 = vec4 && (jobs[i].out_dim % 4 == 0);
  if (vec4) {
    ColJobs Cj = {};
    for (int i = 0; i < njobs; ++i) { Cj.W[i] = J.W[i]; Cj.s[i] = J.s[i]; Cj.G[i] = J.G[i]; Cj.out[i] = J.cbuf[i]; Cj.in_dim[i] = J.in_dim[i]; Cj.out_dim[i] = J.out_dim[i]; }
    hipLaunchKernelGGL(modfc_colsum4_batch_kernel<true>, dim3((max_out + 63) / 64, B, njobs), dim3(256), 0, st, Cj, B, 0.f);
    if (max_out <= 512)
      hipLaunchKernelGGL(modfc_prep_bwd_ws_batch_kernel<2>, dim3((max_in + 3) / 4, njobs), dim3(256), 0, st, J, B);
    else
      hipLaunchKernelGGL(modfc_prep_bwd_ws_batch_kernel<4>, dim3((max_in + 3) / 4, njobs), dim3(256), 0, st, J, B);
    return CIPS_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(modfc_prep_bwd_c_batch_kernel, dim3((max_out + 31) / 32, B, njobs), dim3(256), 0, st, J);
  hipLaunchKernelGGL(modfc_prep_bwd_w_batch_kernel, dim3((unsigned)((max_nw + 255) / 256), njobs), dim3(256), 0, st, J, B);
  hipLaunchKernelGGL(modfc_prep_bwd_s_batch_kernel, dim3((unsigned)(((long long)B * max_in + 3) / 4), njobs), dim3(256), 0, st, J, B);
  return CIPS_CHECK_LAUNCH();
}
