// small_ops.hip — the O(batch) linear layers around the hot path as a handful of grouped launches for gfx950.
//
// Every per-image vector the kernels consume is a Linear of a style vector: the 18 SinStyleMod.modulation layers of the
// CIPS head (exp/comm/models/mod_conv_fc.py:433-436, :474: s = modulation(style), 512 -> 512 / 32) and the gain_fc /
// bias_fc pairs of the three FiLM layers (exp/comm/models/film_layer.py:59-63, :88-93, 128 -> 128 / 64).  As torch ops
// they are 24 forward GEMMs of 32 rows (hipBLASLt, ~11 us each, launch-bound) and 48 backward GEMMs plus bias
// reductions per step — 0.9 ms of a 20 ms step for 0.2 GFLOP.  Here: one launch forward, three backward, over a table of
// jobs that share the batch size; weight-read bound (18 MB), deterministic (fixed summation order, no atomics).
#include "common.h"
#include "../../include/cips3d_hip.h"

namespace {

constexpr int GL_MAX = 32;     // jobs per launch
constexpr int GL_ROWS = 32;    // batch rows per register tile

struct GLJobs {
  const float* x[GL_MAX]; const float* w[GL_MAX]; const float* bias[GL_MAX]; float* y[GL_MAX];
  const float* dy[GL_MAX]; float* dw[GL_MAX]; float* db[GL_MAX];
  int in_dim[GL_MAX], out_dim[GL_MAX];
  int tile0[GL_MAX + 1];       // first workgroup tile of each job (prefix sums)
  int njobs;
};

__device__ __forceinline__ int find_job(const GLJobs& J, int tile) {
  int j = 0;
  while (j + 1 < J.njobs && tile >= J.tile0[j + 1]) ++j;
  return j;
}

// y[b][o] = sum_i x[b][i] w[o][i] + bias[o].  One wave per 4 outputs: the lanes stride over i with 16-byte loads of the
// weight row (coalesced) and of the x rows (LDS), 32 batch rows in registers, then a transpose-reduce over the lanes.
__global__ __launch_bounds__(256) void glin_fwd_kernel(GLJobs J, int B, int opw) {      // opw = outputs per wave (4 or 1)
  extern __shared__ __attribute__((aligned(16))) float xs[];       // [GL_ROWS][in_dim]
  const int job = find_job(J, blockIdx.x);
  const int in_dim = J.in_dim[job], out_dim = J.out_dim[job];
  const int o0 = (blockIdx.x - J.tile0[job]) * 4 * opw;             // 4 waves x opw outputs per workgroup
  const float* __restrict__ x = J.x[job];
  const float* __restrict__ w = J.w[job];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int b0 = 0; b0 < B; b0 += GL_ROWS) {
    const int nb = min(GL_ROWS, B - b0);
    __syncthreads();
    for (int i = threadIdx.x; i < GL_ROWS * in_dim / 4; i += 256) {
      const int r = i / (in_dim / 4), c = i - r * (in_dim / 4);
      reinterpret_cast<float4*>(xs)[i] = r < nb ? reinterpret_cast<const float4*>(x + (long long)(b0 + r) * in_dim)[c]
                                                : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    for (int oo = 0; oo < opw; ++oo) {
      const int o = o0 + wave * opw + oo;
      if (o >= out_dim) break;
      float acc[GL_ROWS];
#pragma unroll
      for (int r = 0; r < GL_ROWS; ++r) acc[r] = 0.f;
      for (int i4 = lane; i4 < in_dim / 4; i4 += 64) {
        const float4 wv = reinterpret_cast<const float4*>(w + (long long)o * in_dim)[i4];
#pragma unroll
        for (int r = 0; r < GL_ROWS; ++r) {
          const float4 xv = reinterpret_cast<const float4*>(xs + r * in_dim)[i4];
          acc[r] = fmaf(wv.x, xv.x, fmaf(wv.y, xv.y, fmaf(wv.z, xv.z, fmaf(wv.w, xv.w, acc[r]))));
        }
      }
      // transpose-reduce: after the step with mask m a lane keeps the rows whose bit matches its own lane bit
      float v16[16], v8[8], v4[4], v2[2], v1;
      {
        const bool up = lane & 32;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const float send = up ? acc[k] : acc[k + 16], keep = up ? acc[k + 16] : acc[k];
          v16[k] = keep + __shfl_xor(send, 32);
        }
      }
      {
        const bool up = lane & 16;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float send = up ? v16[k] : v16[k + 8], keep = up ? v16[k + 8] : v16[k];
          v8[k] = keep + __shfl_xor(send, 16);
        }
      }
      {
        const bool up = lane & 8;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float send = up ? v8[k] : v8[k + 4], keep = up ? v8[k + 4] : v8[k];
          v4[k] = keep + __shfl_xor(send, 8);
        }
      }
      {
        const bool up = lane & 4;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const float send = up ? v4[k] : v4[k + 2], keep = up ? v4[k + 2] : v4[k];
          v2[k] = keep + __shfl_xor(send, 4);
        }
      }
      {
        const bool up = lane & 2;
        const float send = up ? v2[0] : v2[1], keep = up ? v2[1] : v2[0];
        v1 = keep + __shfl_xor(send, 2);
      }
      v1 += __shfl_xor(v1, 1);
      // the lane now holds row r = 16*b5 + 8*b4 + 4*b3 + 2*b2 + b1 (its lane bits 5..1)
      const int r = ((lane >> 5) & 1) * 16 + ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
      if ((lane & 1) == 0 && r < nb)
        J.y[job][(long long)(b0 + r) * out_dim + o] = v1 + (J.bias[job] ? J.bias[job][o] : 0.f);
    }
  }
}

// dw[o][i] = sum_b dy[b][o] x[b][i];  db[o] = sum_b dy[b][o].  Thread = one 16-byte column group of x; the workgroup's 4 outputs
// share every x row it reads; dy[b][o] is uniform over the workgroup and comes through scalar loads.  No LDS and a small register
// footprint on purpose (round 6): the mapping networks' backward runs on a side stream next to the fused SIREN backward, which
// owns the whole LDS of every CU — a kernel like this one still finds a wave slot beside it (see modfc.hip, co-resident forms).
template <int OT>      // outputs per workgroup
__global__ __launch_bounds__(128) void glin_bwd_w_kernel(GLJobs J, int B) {
  const int job = find_job(J, blockIdx.x);
  const int in_dim = J.in_dim[job], out_dim = J.out_dim[job];
  const int o0 = (blockIdx.x - J.tile0[job]) * OT;
  const float* __restrict__ x = J.x[job];
  const float* __restrict__ dy = J.dy[job] + o0;
  const int i4 = threadIdx.x;                                       // in_dim / 4 <= 128
  const bool on = i4 < in_dim / 4;
  float4 acc[OT];
  float sb[OT];
#pragma unroll
  for (int oo = 0; oo < OT; ++oo) { acc[oo] = make_float4(0.f, 0.f, 0.f, 0.f); sb[oo] = 0.f; }
#pragma unroll 2
  for (int b = 0; b < B; ++b) {
    const float4 xv = on ? reinterpret_cast<const float4*>(x + (long long)b * in_dim)[i4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int oo = 0; oo < OT; ++oo) {
      const float g = (o0 + oo < out_dim) ? dy[(long long)b * out_dim + oo] : 0.f;
      sb[oo] += g;
      acc[oo].x = fmaf(g, xv.x, acc[oo].x); acc[oo].y = fmaf(g, xv.y, acc[oo].y);
      acc[oo].z = fmaf(g, xv.z, acc[oo].z); acc[oo].w = fmaf(g, xv.w, acc[oo].w);
    }
  }
#pragma unroll
  for (int oo = 0; oo < OT; ++oo) {
    const int o = o0 + oo;
    if (o < out_dim) {
      if (on) reinterpret_cast<float4*>(J.dw[job] + (long long)o * in_dim)[i4] = acc[oo];
      if (threadIdx.x == 0 && J.db[job]) J.db[job][o] = sb[oo];
    }
  }
}

// dx partials: part[chunk][b][i] = sum over the chunk's (job, o) pairs of dy[b][o] w[o][i]; chunk = OC outputs of one job.
// Workgroup = (chunk, group of 4 batch rows): thread = 4 consecutive i, 4 rows in registers, dy through scalar loads, the chunk's
// weight rows streamed once per row group (L2).  All jobs of a launch share x (same in_dim).  No LDS (see glin_bwd_w_kernel).
constexpr int GLX_ROWS = 4;
template <int OC>      // outputs per chunk: 64, or 16 for launches of few chunks
__global__ __launch_bounds__(128) void glin_bwd_x_kernel(GLJobs J, int B, int in_dim, float* __restrict__ part) {
  const int job = find_job(J, blockIdx.x);
  const int out_dim = J.out_dim[job];
  const int o0 = (blockIdx.x - J.tile0[job]) * OC;
  const int b0 = blockIdx.y * GLX_ROWS;
  const float* __restrict__ w = J.w[job] + (long long)o0 * in_dim;
  const float* __restrict__ dy = J.dy[job] + o0;
  const int i4 = threadIdx.x;
  const bool on = i4 < in_dim / 4;
  float4 acc[GLX_ROWS];
#pragma unroll
  for (int r = 0; r < GLX_ROWS; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int oend = min(OC, out_dim - o0);
#pragma unroll 2
  for (int oo = 0; oo < oend; ++oo) {
    const float4 wv = on ? reinterpret_cast<const float4*>(w + (long long)oo * in_dim)[i4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int r = 0; r < GLX_ROWS; ++r) {
      const float g = (b0 + r < B) ? dy[(long long)(b0 + r) * out_dim + oo] : 0.f;
      acc[r].x = fmaf(g, wv.x, acc[r].x); acc[r].y = fmaf(g, wv.y, acc[r].y);
      acc[r].z = fmaf(g, wv.z, acc[r].z); acc[r].w = fmaf(g, wv.w, acc[r].w);
    }
  }
  if (on) {
#pragma unroll
    for (int r = 0; r < GLX_ROWS; ++r)
      if (b0 + r < B) reinterpret_cast<float4*>(part + ((long long)blockIdx.x * B + b0 + r) * in_dim)[i4] = acc[r];
  }
}

__global__ __launch_bounds__(256) void glin_sum_kernel(const float* __restrict__ part, float* __restrict__ dx, int nchunks, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int c = 0; c < nchunks; ++c) s += part[(long long)c * n + i];
  dx[i] = s;
}

int fill(GLJobs& J, const cips_glin_job* jobs, int njobs, int outs_per_tile) {
  if (!jobs || njobs <= 0 || njobs > GL_MAX) return (int)hipErrorInvalidValue;
  J.njobs = njobs;
  int t = 0;
  for (int j = 0; j < njobs; ++j) {
    const cips_glin_job& q = jobs[j];
    if (!q.x || !q.w || q.in_dim <= 0 || q.out_dim <= 0 || (q.in_dim & 3) || q.in_dim > 512) return (int)hipErrorInvalidValue;
    J.x[j] = q.x; J.w[j] = q.w; J.bias[j] = q.bias; J.y[j] = q.y; J.dy[j] = q.dy; J.dw[j] = q.dw; J.db[j] = q.db;
    J.in_dim[j] = q.in_dim; J.out_dim[j] = q.out_dim;
    J.tile0[j] = t;
    t += (q.out_dim + outs_per_tile - 1) / outs_per_tile;
  }
  J.tile0[njobs] = t;
  return 0;
}


// ------------------------------------------------------------------------------------------------------------------
// Row-wise normalisation + activation of the two z -> style mapping MLPs (exp/cips3d/models/multi_head_mapping.py:13-19
// PixelNorm, :62-84 [Linear, LayerNorm, LeakyReLU(0.2)] x L): one workgroup per batch row, the row in registers.
//   mode bit 0: LayerNorm (biased variance, eps 1e-5, affine gamma / beta) | bit 1: LeakyReLU(slope) after it
//   mode bit 2: PixelNorm  y = x * rsqrt(mean(x^2) + 1e-8)  (no affine, no activation)
// stats (rows, 2): [mean, rstd] (LayerNorm) or [unused, r] (PixelNorm), kept for the backward.
constexpr int RN_MAXC = 1024;

__device__ __forceinline__ float rn_block_sum(float v, float* red) {
  const int t = threadIdx.x;
  red[t] = v;
  __syncthreads();
  for (int s_ = 128; s_ > 0; s_ >>= 1) {
    if (t < s_) red[t] += red[t + s_];
    __syncthreads();
  }
  const float r = red[0];
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(256) void rownorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ y,
                                                          float* __restrict__ stats, int cols, int mode, float slope) {
  __shared__ float red[256];
  const int row = blockIdx.x, t = threadIdx.x;
  const float* xr = x + (long long)row * cols;
  float v[RN_MAXC / 256];
#pragma unroll
  for (int q = 0; q < RN_MAXC / 256; ++q) v[q] = (t + 256 * q < cols) ? xr[t + 256 * q] : 0.f;
  float out[RN_MAXC / 256];
  if (mode & 4) {
    float ss = 0.f;
#pragma unroll
    for (int q = 0; q < RN_MAXC / 256; ++q) ss = fmaf(v[q], v[q], ss);
    const float r = rsqrtf(rn_block_sum(ss, red) / (float)cols + 1e-8f);
#pragma unroll
    for (int q = 0; q < RN_MAXC / 256; ++q) out[q] = v[q] * r;
    if (t == 0) { stats[2 * row] = 0.f; stats[2 * row + 1] = r; }
  } else {
    float mean = 0.f, rstd = 1.f;
    if (mode & 1) {
      float s1 = 0.f;
#pragma unroll
      for (int q = 0; q < RN_MAXC / 256; ++q) s1 += v[q];
      mean = rn_block_sum(s1, red) / (float)cols;
      float s2 = 0.f;
#pragma unroll
      for (int q = 0; q < RN_MAXC / 256; ++q) { const float dlt = (t + 256 * q < cols) ? v[q] - mean : 0.f; s2 = fmaf(dlt, dlt, s2); }
      rstd = rsqrtf(rn_block_sum(s2, red) / (float)cols + 1e-5f);
    }
#pragma unroll
    for (int q = 0; q < RN_MAXC / 256; ++q) {
      const int c = t + 256 * q;
      float o = v[q];
      if ((mode & 1) && c < cols) o = (v[q] - mean) * rstd * gamma[c] + beta[c];
      if (mode & 2) o = o > 0.f ? o : o * slope;
      out[q] = o;
    }
    if (t == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
  }
#pragma unroll
  for (int q = 0; q < RN_MAXC / 256; ++q)
    if (t + 256 * q < cols) y[(long long)row * cols + t + 256 * q] = out[q];
}

// dx for one row; dyhat (rows, cols) = dL/d(gamma * xhat + beta) is written for the column reductions of d gamma / d beta.
// One WAVE per row, two passes over the row (the second one re-reads it from L2), reductions by DPP: no LDS, ~20 registers
// (see glin_bwd_w_kernel for why).
__global__ __launch_bounds__(64) void rownorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ yout,
                                                         const float* __restrict__ gamma, const float* __restrict__ stats,
                                                         const float* __restrict__ dy, float* __restrict__ dx,
                                                         float* __restrict__ dyhat, int cols, int mode, float slope) {
  const int row = blockIdx.x, t = threadIdx.x;
  const long long base = (long long)row * cols;
  auto gate = [&](int c) {                      // upstream gradient through the activation (gate from the stored output: same sign)
    float gg = dy[base + c];
    if (mode & 2) gg *= (yout[base + c] > 0.f) ? 1.f : slope;
    return gg;
  };
  if (mode & 4) {                     // y = x r: dx = r (dy - y mean(dy y))
    const float r = stats[2 * row + 1];
    float s = 0.f;
    for (int c = t; c < cols; c += 64) s = fmaf(gate(c), x[base + c] * r, s);
    const float m = wave_sum_dpp(s) / (float)cols;
    for (int c = t; c < cols; c += 64) dx[base + c] = r * (gate(c) - x[base + c] * r * m);
    return;
  }
  if (!(mode & 1)) {                  // activation only
    for (int c = t; c < cols; c += 64) dx[base + c] = gate(c);
    return;
  }
  const float mean = stats[2 * row], rstd = stats[2 * row + 1];
  float s1 = 0.f, s2 = 0.f;
  for (int c = t; c < cols; c += 64) {
    const float g = gate(c);
    const float gx = g * gamma[c];
    dyhat[base + c] = g;
    s1 += gx;
    s2 = fmaf(gx, (x[base + c] - mean) * rstd, s2);
  }
  const float m1 = wave_sum_dpp(s1) / (float)cols;
  const float m2 = wave_sum_dpp(s2) / (float)cols;
  for (int c = t; c < cols; c += 64) {
    const float gx = dyhat[base + c] * gamma[c];
    dx[base + c] = rstd * (gx - m1 - (x[base + c] - mean) * rstd * m2);
  }
}

// d gamma[c] = sum_rows dyhat * xhat, d beta[c] = sum_rows dyhat  (fixed order)
__global__ __launch_bounds__(256) void rownorm_bwd_affine_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                                 const float* __restrict__ dyhat, float* __restrict__ dgamma,
                                                                 float* __restrict__ dbeta, int rows, int cols) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  float a = 0.f, b = 0.f;
  for (int r = 0; r < rows; ++r) {
    const float g = dyhat[(long long)r * cols + c];
    a = fmaf(g, (x[(long long)r * cols + c] - stats[2 * r]) * stats[2 * r + 1], a);
    b += g;
  }
  dgamma[c] = a; dbeta[c] = b;
}


// ------------------------------------------------------------------------------------------------------------------
// Camera pose of a training batch in ONE launch (exp/comm/comm_utils.py:451-581: sample_camera_positions for the
// 'gaussian' / 'normal' / 'uniform' distributions + create_cam2world_matrix with the (0, 1, 0) up vector): from the raw
// draws to pitch, yaw, the camera origin and cam2world.  As torch ops this is ~45 launches of one-wave kernels on (b, 1)
// and (b, 3) tensors per step, each a ~5 us node of the captured graph.  The arithmetic follows the reference's operation
// order with separately rounded multiplies / adds (no fma contraction), so the result agrees with the op-by-op form to
// the last bits of sinf / cosf.
__global__ __launch_bounds__(64) void camera_pose_kernel(const float* __restrict__ th_raw, const float* __restrict__ ph_raw,
                                                         int uniform, float hs, float hm, float vs, float vm,
                                                         float phi_lo, float phi_hi, float* __restrict__ pitch_yaw,
                                                         float* __restrict__ origin, float* __restrict__ c2w, int B) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= B) return;
  float th, ph;
  if (uniform) {                      // (u - 0.5) * 2 * stddev + mean
    th = __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(th_raw[i], 0.5f), 2.f), hs), hm);
    ph = __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(ph_raw[i], 0.5f), 2.f), vs), vm);
  } else {                            // g * stddev + mean
    th = __fadd_rn(__fmul_rn(th_raw[i], hs), hm);
    ph = __fadd_rn(__fmul_rn(ph_raw[i], vs), vm);
  }
  ph = fminf(fmaxf(ph, phi_lo), phi_hi);
  const float sp = sinf(ph), cp = cosf(ph), st = sinf(th), ct = cosf(th);
  const float ox = __fmul_rn(sp, ct), oy = cp, oz = __fmul_rn(sp, st);          // r = 1
  auto norm3 = [](float x, float y, float z) { return sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z))); };
  // forward = normalize(-origin), normalised once more inside create_cam2world_matrix
  float n = norm3(-ox, -oy, -oz);
  float fx = -ox / n, fy = -oy / n, fz = -oz / n;
  n = norm3(fx, fy, fz);
  fx /= n; fy /= n; fz /= n;
  // left = normalize(cross((0,1,0), f)) = normalize((f_z, 0, -f_x)); up = normalize(cross(f, left))
  float lx = fz, ly = 0.f, lz = -fx;
  n = norm3(lx, ly, lz);
  lx /= n; ly /= n; lz /= n;
  float ux = __fsub_rn(__fmul_rn(fy, lz), __fmul_rn(fz, ly));
  float uy = __fsub_rn(__fmul_rn(fz, lx), __fmul_rn(fx, lz));
  float uz = __fsub_rn(__fmul_rn(fx, ly), __fmul_rn(fy, lx));
  n = norm3(ux, uy, uz);
  ux /= n; uy /= n; uz /= n;
  pitch_yaw[2 * i] = ph; pitch_yaw[2 * i + 1] = th;
  origin[3 * i] = ox; origin[3 * i + 1] = oy; origin[3 * i + 2] = oz;
  float* m = c2w + 16 * i;            // trans @ rot: columns (-left, up, -forward, origin)
  m[0] = -lx; m[1] = ux; m[2] = -fx; m[3] = ox;
  m[4] = -ly; m[5] = uy; m[6] = -fy; m[7] = oy;
  m[8] = -lz; m[9] = uz; m[10] = -fz; m[11] = oz;
  m[12] = 0.f; m[13] = 0.f; m[14] = 0.f; m[15] = 1.f;
}

}  // namespace

extern "C" int cips_grouped_linear_max_jobs(void) { return GL_MAX; }

extern "C" int cips_grouped_linear_fwd(const cips_glin_job* jobs, int njobs, int B, cips_stream_t stream) {
  GLJobs J;
  int rc = fill(J, jobs, njobs, 16);
  if (rc) return rc;
  if (B <= 0) return (int)hipErrorInvalidValue;
  int opw = 4;
  if (J.tile0[njobs] < 128) { opw = 1; rc = fill(J, jobs, njobs, 4); if (rc) return rc; }     // few outputs: 4 per workgroup
  int max_in = 0;
  for (int j = 0; j < njobs; ++j) { if (!jobs[j].y) return (int)hipErrorInvalidValue; max_in = jobs[j].in_dim > max_in ? jobs[j].in_dim : max_in; }
  const size_t smem = (size_t)GL_ROWS * max_in * sizeof(float);
  static bool attr_set = false;
  CIPS_PER_DEVICE(attr_set, false);
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)glin_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GL_ROWS * 512 * 4); attr_set = true; }
  hipLaunchKernelGGL(glin_fwd_kernel, dim3(J.tile0[njobs]), dim3(256), smem, (hipStream_t)stream, J, B, opw);
  return CIPS_CHECK_LAUNCH();
}

extern "C" int cips_grouped_linear_bwd(const cips_glin_job* jobs, int njobs, int B, float* dx, float* scratch,
                                       long long scratch_floats, cips_stream_t stream) {
  GLJobs J;
  int rc = fill(J, jobs, njobs, 16);
  if (rc) return rc;
  if (B <= 0) return (int)hipErrorInvalidValue;
  for (int j = 0; j < njobs; ++j) if (!jobs[j].dy || !jobs[j].dw) return (int)hipErrorInvalidValue;
  rc = fill(J, jobs, njobs, 4);
  if (rc) return rc;
  hipLaunchKernelGGL(glin_bwd_w_kernel<4>, dim3(J.tile0[njobs]), dim3(128), 0, (hipStream_t)stream, J, B);
  if (dx) {
    const int in_dim = jobs[0].in_dim;
    for (int j = 1; j < njobs; ++j) if (jobs[j].in_dim != in_dim || jobs[j].x != jobs[0].x) return (int)hipErrorInvalidValue;
    GLJobs K;
    rc = fill(K, jobs, njobs, 64);
    if (rc) return rc;
    const bool fine = K.tile0[njobs] < 64;
    if (fine) { rc = fill(K, jobs, njobs, 16); if (rc) return rc; }
    const int nchunks = K.tile0[njobs];
    if (!scratch || scratch_floats < (long long)nchunks * B * in_dim) return (int)hipErrorInvalidValue;
    const dim3 gx(nchunks, (B + GLX_ROWS - 1) / GLX_ROWS);
    if (fine) hipLaunchKernelGGL(glin_bwd_x_kernel<16>, gx, dim3(128), 0, (hipStream_t)stream, K, B, in_dim, scratch);
    else hipLaunchKernelGGL(glin_bwd_x_kernel<64>, gx, dim3(128), 0, (hipStream_t)stream, K, B, in_dim, scratch);
    const long long n = (long long)B * in_dim;
    hipLaunchKernelGGL(glin_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, scratch, dx, nchunks, n);
  }
  return CIPS_CHECK_LAUNCH();
}

extern "C" long long cips_grouped_linear_scratch(const cips_glin_job* jobs, int njobs, int B) {
  if (!jobs || njobs <= 0) return 0;
  long long chunks = 0;
  for (int j = 0; j < njobs; ++j) chunks += (jobs[j].out_dim + 63) / 64;
  if (chunks < 64) {            // the finer chunking of cips_grouped_linear_bwd
    chunks = 0;
    for (int j = 0; j < njobs; ++j) chunks += (jobs[j].out_dim + 15) / 16;
  }
  return chunks * B * jobs[0].in_dim;
}

extern "C" int cips_rownorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stats, int rows,
                                int cols, int mode, float slope, cips_stream_t stream) {
  if (!x || !y || !stats || rows <= 0 || cols <= 0 || cols > RN_MAXC || ((mode & 1) && (!gamma || !beta)) || ((mode & 4) && (mode & 3)))
    return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(rownorm_fwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, stats, cols, mode, slope);
  return CIPS_CHECK_LAUNCH();
}

extern "C" int cips_rownorm_bwd(const float* x, const float* y, const float* gamma, const float* stats, const float* dy,
                                float* dx, float* dyhat, float* dgamma, float* dbeta, int rows, int cols, int mode, float slope,
                                cips_stream_t stream) {
  if (!x || !y || !stats || !dy || !dx || rows <= 0 || cols <= 0 || cols > RN_MAXC || ((mode & 4) && (mode & 3))) return (int)hipErrorInvalidValue;
  if ((mode & 1) && (!gamma || !dyhat || !dgamma || !dbeta)) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(rownorm_bwd_kernel, dim3(rows), dim3(64), 0, st, x, y, gamma, stats, dy, dx, dyhat, cols, mode, slope);
  if (mode & 1)
    hipLaunchKernelGGL(rownorm_bwd_affine_kernel, dim3((cols + 255) / 256), dim3(256), 0, st, x, stats, dyhat, dgamma, dbeta, rows, cols);
  return CIPS_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------------------------
// EqualLinear (exp/cips3d/models/discriminator.py:254-288: F.linear(input, weight * scale) [+ bias * lr_mul]) and the two
// other bilinear forms its autograd needs; each form's own gradients are the other two, which closes the
// double-backward of the R1 penalty (train.py:387-394):
//   mode 0   y  (B, O) = s * x (B, K) . w^T (O, K)  [+ bias (O) * bias_scale]
//   mode 1   dx (B, K) = s * g (B, O) . w (O, K)
//   mode 2   dw (O, K) = s * g^T (O, B) . x (B, K)
// The 8192 -> 512 layer (final_conv features -> space_linear) runs on the exact-fp32 MFMA GEMM (gemm_f32.hip); its
// forward contracts over 8192 inputs for at most a few dozen rows, i.e. four 128x128 output tiles: the contraction is
// cut into 512-wide chunks (the GEMM's batch dimension) and the partial products are summed in chunk order here.  The
// 512 -> 1 layer has no dimension a matrix tile could hold: three streaming kernels.
namespace {
__global__ __launch_bounds__(256) void eql_sum_chunks_kernel(const float* __restrict__ part, float* __restrict__ y, int nch, int n,
                                                             int O, const float* __restrict__ bias, float bias_scale) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    float v = part[i];
    for (int c = 1; c < nch; ++c) v += part[(long long)c * n + i];
    if (bias) v += bias[i % O] * bias_scale;
    y[i] = v;
  }
}
// mode 0, narrow output: one workgroup per row b, every thread a strided slice of k, tree sum in LDS
__global__ __launch_bounds__(256) void eql_fwd_small_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float bias_scale, float s,
                                                            float* __restrict__ y, int K, int O) {
  __shared__ float red[256];
  const int b = blockIdx.x;
  for (int o = 0; o < O; ++o) {
    float t = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) t = fmaf(x[(long long)b * K + k], w[(long long)o * K + k], t);
    red[threadIdx.x] = t;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
      if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
      __syncthreads();
    }
    if (threadIdx.x == 0) y[(long long)b * O + o] = s * red[0] + (bias ? bias[o] * bias_scale : 0.f);
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void eql_dx_small_kernel(const float* __restrict__ g, const float* __restrict__ w, float s,
                                                           float* __restrict__ dx, int B, int K, int O) {
  const long long n = (long long)B * K;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int b = (int)(i / K), k = (int)(i - (long long)b * K);
    float t = 0.f;
    for (int o = 0; o < O; ++o) t = fmaf(g[(long long)b * O + o], w[(long long)o * K + k], t);
    dx[i] = s * t;
  }
}
__global__ __launch_bounds__(256) void eql_dw_small_kernel(const float* __restrict__ g, const float* __restrict__ x, float s,
                                                           float* __restrict__ dw, int B, int K, int O) {
  const long long n = (long long)O * K;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int o = (int)(i / K), k = (int)(i - (long long)o * K);
    float t = 0.f;
    for (int b = 0; b < B; ++b) t = fmaf(g[(long long)b * O + o], x[(long long)b * K + k], t);
    dw[i] = s * t;
  }
}
__host__ int eql_chunks(int K) { return (K >= 2048 && K % 512 == 0) ? K / 512 : 1; }
}  // namespace

extern "C" long long cips_equal_linear_scratch(int mode, int B, int K, int O) {
  if (mode != 0 || (O & 3) || (K & 3)) return 0;
  const int nch = eql_chunks(K);
  return nch > 1 ? (long long)nch * B * O : 0;
}

extern "C" int cips_equal_linear(int mode, const float* a, const float* b, const float* bias, float bias_scale, float s,
                                 float* out, float* scratch, int B, int K, int O, cips_stream_t stream) {
  if (!a || !b || !out || B <= 0 || K <= 0 || O <= 0 || mode < 0 || mode > 2) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  const bool wide = !(O & 3) && !(K & 3);
  if (!wide) {
    if (mode == 0) hipLaunchKernelGGL(eql_fwd_small_kernel, dim3(B), dim3(256), 0, st, a, b, bias, bias_scale, s, out, K, O);
    else {
      const long long n = mode == 1 ? (long long)B * K : (long long)O * K;
      const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
      if (mode == 1) hipLaunchKernelGGL(eql_dx_small_kernel, dim3(blocks), dim3(256), 0, st, a, b, s, out, B, K, O);
      else hipLaunchKernelGGL(eql_dw_small_kernel, dim3(blocks), dim3(256), 0, st, a, b, s, out, B, K, O);
    }
    return CIPS_CHECK_LAUNCH();
  }
  cips_gemm_desc d = {};
  d.alpha = s; d.batch = 1;
  if (mode == 0) {                 // y = s x w^T: A = x (B, K), B = w stored (O, K) ("NT")
    const int nch = eql_chunks(K);
    if (nch > 1 && !scratch) return (int)hipErrorInvalidValue;
    d.A = a; d.B = b; d.C = nch > 1 ? scratch : out;
    d.M = B; d.N = O; d.K = K / nch; d.lda = K; d.ldb = K; d.ldc = O; d.b_nmajor = 1;
    d.batch = nch; d.strideA = K / nch; d.strideB = K / nch; d.strideC = (long long)B * O;
    if (nch == 1 && bias) { /* bias added by the finishing pass below */ }
    int rc = cips_gemm_f32(&d, stream);
    if (rc) return rc;
    if (nch > 1 || bias) {
      const int n = B * O;
      hipLaunchKernelGGL(eql_sum_chunks_kernel, dim3((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024), dim3(256), 0, st,
                         nch > 1 ? scratch : out, out, nch, n, O, bias, bias_scale);
    }
    return CIPS_CHECK_LAUNCH();
  }
  if (mode == 1) {                 // dx = s g w: A = g (B, O), B = w (O, K) row-major
    d.A = a; d.B = b; d.C = out; d.M = B; d.N = K; d.K = O; d.lda = O; d.ldb = K; d.ldc = K;
  } else {                         // dw = s g^T x: A = g stored (B, O) = (contraction, M) ("TN"), B = x (B, K)
    d.A = a; d.B = b; d.C = out; d.M = O; d.N = K; d.K = B; d.lda = O; d.ldb = K; d.ldc = K; d.a_kmajor = 1;
  }
  return cips_gemm_f32(&d, stream);
}

extern "C" int cips_camera_pose(const float* theta_raw, const float* phi_raw, int uniform, float h_stddev, float h_mean,
                                float v_stddev, float v_mean, float* pitch_yaw, float* origin, float* cam2world, int B,
                                cips_stream_t stream) {
  if (!theta_raw || !phi_raw || !pitch_yaw || !origin || !cam2world || B <= 0) return (int)hipErrorInvalidValue;
  const float lo = 1e-5f, hi = (float)(3.14159265358979323846 - 1e-5);
  hipLaunchKernelGGL(camera_pose_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, theta_raw, phi_raw, uniform,
                     h_stddev, h_mean, v_stddev, v_mean, lo, hi, pitch_yaw, origin, cam2world, B);
  return CIPS_CHECK_LAUNCH();
}
