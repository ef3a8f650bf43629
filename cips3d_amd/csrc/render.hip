// render.hip — ray set-up, hierarchical resampling and the merge + alpha-composite
// (fancy_integration) forward / backward for gfx950.  All of these are HBM-bandwidth bound
// streaming kernels: no MFMA, coalesced 128-byte rows, per-ray scans done by
// an 8-lane segment of a wave64 (8 rays per wave; the 8 lanes of a segment cover the 32
// feature channels of one sample with one float4 each, i.e. one full 128 B line per sample).
#include "common.h"
#include "../../include/cips3d_hip.h"
#include "raygen.h"

namespace {

// ------------------------------------------------------------------------------------
// H1: rays.  Follows exp/comm/comm_utils.py:365-412 (get_initial_rays_trig), :416-438
// (perturb_points) and the three bmm's of :584-679 (transform_sampled_points).
// One thread per sample point; everything is recomputed from the pixel index.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rays_kernel(const float* __restrict__ xg, const float* __restrict__ yg,
                                                   const float* __restrict__ zg, float zc,
                                                   const float* __restrict__ c2w, const float* __restrict__ jitter,
                                                   float* __restrict__ points, float* __restrict__ zout,
                                                   float* __restrict__ dirs, int B, int H, int W, int S) {
  const long long total = (long long)B * H * W * S;
  const int n = H * W;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(idx % S);
    const long long rr = idx / S;
    const int ray = (int)(rr % n);
    const int b = (int)(rr / n);
    const int row = ray / W, col = ray % W;
    const float x = xg[col], y = yg[row];
    const float nrm = sqrtf(x * x + y * y + zc * zc);
    const float dx = x / nrm, dy = y / nrm, dz = zc / nrm;
    float z = zg[i];
    float px = dx * z, py = dy * z, pz = dz * z;
    if (jitter) {
      const float off = (jitter[idx] - 0.5f) * (zg[1] - zg[0]);
      z = z + off;
      px = px + off * dx; py = py + off * dy; pz = pz + off * dz;
    }
    const float* M = c2w + (long long)b * 16;
    points[idx * 3 + 0] = ((M[0] * px + M[1] * py) + M[2] * pz) + M[3];
    points[idx * 3 + 1] = ((M[4] * px + M[5] * py) + M[6] * pz) + M[7];
    points[idx * 3 + 2] = ((M[8] * px + M[9] * py) + M[10] * pz) + M[11];
    zout[idx] = z;
    if (i == 0) {
      dirs[rr * 3 + 0] = (M[0] * dx + M[1] * dy) + M[2] * dz;
      dirs[rr * 3 + 1] = (M[4] * dx + M[5] * dy) + M[6] * dz;
      dirs[rr * 3 + 2] = (M[8] * dx + M[9] * dy) + M[10] * dz;
    }
  }
}

__device__ __forceinline__ float clamp_density(float x, int mode) {
  if (mode == 1) return (x > 20.f) ? x : log1pf(expf(x));  // F.softplus (beta=1, threshold=20)
  return fmaxf(x, 0.f);                                   // F.relu
}
__device__ __forceinline__ float clamp_density_grad(float x, int mode) {
  if (mode == 1) return (x > 20.f) ? 1.f : 1.f / (1.f + expf(-x));
  return x > 0.f ? 1.f : 0.f;
}

// ------------------------------------------------------------------------------------
// H3a: coarse weights -> pdf -> cdf -> inverse-CDF samples -> fine points.
// Follows exp/dev/nerf_inr/models/generator_nerf_inr.py:564-592 and
// exp/pigan/pigan_utils.py:181-209 (sample_pdf), :239-258 (weights of fancy_integration).
// ATen's CPU cumprod/cumsum accumulate in double (acc_type<float,false>) and round each
// prefix to float; the per-ray scans here do the same so that the oracle's integer
// bookkeeping (searchsorted indices) is reproduced on identical inputs.
// ------------------------------------------------------------------------------------
struct ResampleArgs {
  const float *sigma, *z, *noise, *u, *origins, *dirs;
  const float* cdf_in;      // optional (R, S-1): use this cdf instead of the one computed here (bookkeeping contract test)
  RayGen rg;                // dirs == NULL: ray directions and origins come from the ray parameters
  int have_rg;
  float noise_std;
  float *fine_z, *fine_pts, *weights_out, *cdf_out;
  long long* inds_out;
  int B, n, S, clamp_mode;
};

constexpr int SEG = 8;  // lanes per ray

__global__ __launch_bounds__(256) void resample_kernel(ResampleArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int S = a.S;
  const int rays_per_block = blockDim.x / SEG;
  const int lr = threadIdx.x / SEG, sub = threadIdx.x % SEG;
  const long long R = (long long)a.B * a.n;
  const long long ray_raw = (long long)blockIdx.x * rays_per_block + lr;
  const bool active = ray_raw < R;           // inactive segments recompute the last ray, stores predicated
  const long long ray = active ? ray_raw : R - 1;
  float* zs = sm + lr * (4 * S);
  float* ws = zs + S;
  float* cdf = ws + S;   // S-1 entries
  float* bins = cdf + S; // S-1 entries

  for (int i = sub; i < S; i += SEG) {
    zs[i] = a.z[ray * S + i];
    float sg = a.sigma[ray * S + i];
    if (a.noise) sg += a.noise[ray * S + i] * a.noise_std;
    ws[i] = sg;
  }
  __syncthreads();
  if (sub == 0) {
    double T = 1.0;
    for (int i = 0; i < S; ++i) {
      const float delta = (i + 1 < S) ? (zs[i + 1] - zs[i]) : 1e10f;
      const float dens = clamp_density(ws[i], a.clamp_mode);
      const float alpha = 1.f - expf(-delta * dens);
      const float Tf = (float)T;
      T *= (double)(1.f - alpha + 1e-10f);
      ws[i] = alpha * Tf;
    }
    // weights[:,1:-1] + 1e-5 (generator_nerf_inr.py:572) + eps (pigan_utils.py:181)
    float sum = 0.f;
    for (int k = 0; k < S - 2; ++k) sum += (ws[k + 1] + 1e-5f) + 1e-5f;
    double acc = 0.0;
    cdf[0] = 0.f;
    for (int k = 0; k < S - 2; ++k) {
      const float pdf = ((ws[k + 1] + 1e-5f) + 1e-5f) / sum;
      acc += (double)pdf;
      cdf[k + 1] = (float)acc;
    }
    for (int j = 0; j < S - 1; ++j) bins[j] = 0.5f * (zs[j] + zs[j + 1]);
    if (a.cdf_in)
      for (int j = 0; j < S - 1; ++j) cdf[j] = a.cdf_in[ray * (S - 1) + j];
  }
  __syncthreads();
  const int b = (int)(ray / a.n);
  float ox, oy, oz, dx, dy, dz;
  if (a.have_rg) {          // same expressions as rays_kernel (directions) / the camera matrix' translation column
    const float* M = a.rg.c2w + (long long)b * 16;
    const RayDir d = ray_dir(a.rg, (int)(ray - (long long)b * a.n));
    dx = (M[0] * d.dx + M[1] * d.dy) + M[2] * d.dz;
    dy = (M[4] * d.dx + M[5] * d.dy) + M[6] * d.dz;
    dz = (M[8] * d.dx + M[9] * d.dy) + M[10] * d.dz;
    ox = M[3]; oy = M[7]; oz = M[11];
  } else {
    ox = a.origins[b * 3 + 0]; oy = a.origins[b * 3 + 1]; oz = a.origins[b * 3 + 2];
    dx = a.dirs[ray * 3 + 0]; dy = a.dirs[ray * 3 + 1]; dz = a.dirs[ray * 3 + 2];
  }
  for (int i = sub; i < S; i += SEG) {
    const float u = a.u[ray * S + i];
    int ind = 0;
    for (int j = 0; j < S - 1; ++j) ind += (cdf[j] < u) ? 1 : 0;  // searchsorted(right=False)
    const int below = max(ind - 1, 0);
    const int above = min(ind, S - 2);
    const float c0 = cdf[below], c1 = cdf[above];
    const float b0 = bins[below], b1 = bins[above];
    float denom = c1 - c0;
    if (denom < 1e-5f) denom = 1.f;
    const float smp = b0 + (u - c0) / denom * (b1 - b0);
    if (!active) continue;
    a.fine_z[ray * S + i] = smp;
    if (a.fine_pts) {
      float* fp = a.fine_pts + (ray * S + i) * 3;
      fp[0] = ox + dx * smp; fp[1] = oy + dy * smp; fp[2] = oz + dz * smp;
    }
    if (a.inds_out) a.inds_out[ray * S + i] = ind;
    if (a.weights_out) a.weights_out[ray * S + i] = ws[i];
  }
  if (a.cdf_out && active)
    for (int j = sub; j < S - 1; j += SEG) a.cdf_out[ray * (S - 1) + j] = cdf[j];
}

// ------------------------------------------------------------------------------------
// H3b: merge + composite.  Follows exp/cips3d/models/generator.py:1733-1752 and
// exp/pigan/pigan_utils.py:212-273.
// ------------------------------------------------------------------------------------
struct CompArgs {
  const float *feat_c, *sig_c, *z_c, *feat_f, *sig_f, *z_f, *noise;
  float noise_std;
  // forward outputs
  float *fea, *depth, *weights, *zsorted;
  int* order;
  // backward
  const int* order_in;
  const float* dfea;
  float *dfeat_c, *dsig_c, *dfeat_f, *dsig_f;
  long long R;
  int S, E, clamp_mode, flags;
  // optional branch masks of the relu clamp per (ray, sorted position), see cips_composite_fwd in cips3d_hip.h:
  // clamp_pin supplies the branch (0 = clamped), clamp_rec receives the branch taken; both NULL in production
  const unsigned char* clamp_pin;
  unsigned char* clamp_rec;
};

// relu(sigma + noise) with the branch taken from a supplied decision (the other branch's linear extension), so that two
// evaluations whose pre-activations differ by rounding differentiate the SAME function (tests only; pigan_utils.py:246-252)
__device__ __forceinline__ bool clamp_pass(const CompArgs& a, float x, long long idx, bool store) {
  bool pass = x > 0.f;
  if (a.clamp_pin) pass = a.clamp_pin[idx] != 0;
  if (a.clamp_rec && store) a.clamp_rec[idx] = pass ? 1 : 0;
  return pass;
}

__device__ __forceinline__ const float* feat_row(const CompArgs& a, long long ray, int i) {
  // i indexes torch.cat([fine, coarse]) when a fine set exists
  if (a.feat_f) return (i < a.S) ? a.feat_f + (ray * a.S + i) * 32 : a.feat_c + (ray * a.S + (i - a.S)) * 32;
  return a.feat_c + (ray * a.S + i) * 32;
}

__global__ __launch_bounds__(256) void composite_fwd_kernel(CompArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int S = a.S, E = a.E;
  const int rays_per_block = blockDim.x / SEG;
  const int lr = threadIdx.x / SEG, sub = threadIdx.x % SEG;
  const long long ray_raw = (long long)blockIdx.x * rays_per_block + lr;
  const bool active = ray_raw < a.R;         // inactive segments recompute the last ray, stores predicated
  const long long ray = active ? ray_raw : a.R - 1;
  float* zall = sm + lr * (4 * E);
  float* sall = zall + E;
  int* ord = reinterpret_cast<int*>(sall + E);

  for (int i = sub; i < E; i += SEG) {
    if (a.feat_f) {
      zall[i] = (i < S) ? a.z_f[ray * S + i] : a.z_c[ray * S + (i - S)];
      sall[i] = (i < S) ? a.sig_f[ray * S + i] : a.sig_c[ray * S + (i - S)];
    } else {
      zall[i] = a.z_c[ray * S + i];
      sall[i] = a.sig_c[ray * S + i];
      ord[i] = i;
    }
  }
  __syncthreads();
  if (a.feat_f) {
    // stable ascending rank (torch.sort ties are measure-zero for continuous z)
    for (int i = sub; i < E; i += SEG) {
      const float zi = zall[i];
      int rank = 0;
      for (int j = 0; j < E; ++j) {
        const float zj = zall[j];
        rank += (zj < zi || (zj == zi && j < i)) ? 1 : 0;
      }
      ord[rank] = i;
    }
  }
  __syncthreads();

  // alpha of every sorted position, in parallel over the segment's lanes (one expf per lane and 8 positions instead
  // of E serial ones): alpha_k = 1 - exp(-delta_k * clamp(sigma_k + noise_k))
  float* al = reinterpret_cast<float*>(ord + E);
  for (int k = sub; k < E; k += SEG) {
    const int i = ord[k];
    const float zk = zall[i];
    const float delta = (k + 1 < E) ? (zall[ord[k + 1]] - zk) : 1e10f;
    float sg = sall[i];
    if (a.noise) sg += a.noise[ray * E + k] * a.noise_std;
    float dens = clamp_density(sg, a.clamp_mode);
    if ((a.clamp_pin || a.clamp_rec) && a.clamp_mode == 0) dens = clamp_pass(a, sg, ray * E + k, active) ? sg : 0.f;
    al[k] = 1.f - expf(-delta * dens);
  }
  __syncthreads();

  float4 F = make_float4(0.f, 0.f, 0.f, 0.f);
  float depth = 0.f, wsum = 0.f, wlast = 0.f, zlast = 0.f;
  float4 flast = make_float4(0.f, 0.f, 0.f, 0.f);
  double T = 1.0;      // transmittance in double like ATen's CPU cumprod (acc_type), rounded to float per prefix
  // feature rows in batches of 8 independent 16-byte loads per lane (one 128-B line per sample and segment), then the
  // in-order accumulation: the serial chain is one double multiply and one fma per sample, no load and no expf in it
  for (int k0 = 0; k0 < E; k0 += 8) {
    float4 f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = (k0 + j < E) ? k0 + j : E - 1;
      f[j] = *reinterpret_cast<const float4*>(feat_row(a, ray, ord[k]) + 4 * sub);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + j;
      if (k < E) {
        const int i = ord[k];
        const float zk = zall[i];
        const float alpha = al[k];
        const float w = alpha * (float)T;
        T *= (double)(1.f - alpha + 1e-10f);
        F.x = fmaf(w, f[j].x, F.x); F.y = fmaf(w, f[j].y, F.y); F.z = fmaf(w, f[j].z, F.z); F.w = fmaf(w, f[j].w, F.w);
        depth = fmaf(w, zk, depth);
        wsum += w;
        if (k == E - 1) { wlast = w; zlast = zk; flast = f[j]; }
        if ((k % SEG) == sub && active) {
          if (a.weights && !(k == E - 1 && (a.flags & 1))) a.weights[ray * E + k] = w;
          if (a.order) a.order[ray * E + k] = i;
          if (a.zsorted) a.zsorted[ray * E + k] = zk;
        }
      }
    }
  }
  if (a.flags & 1) {  // last_back: weights[:, :, -1] += 1 - weights_sum
    const float extra = 1.f - wsum;
    F.x = fmaf(extra, flast.x, F.x); F.y = fmaf(extra, flast.y, F.y);
    F.z = fmaf(extra, flast.z, F.z); F.w = fmaf(extra, flast.w, F.w);
    depth = fmaf(extra, zlast, depth);
    if (a.weights && ((E - 1) % SEG) == sub && active) a.weights[ray * E + E - 1] = wlast + extra;
  }
  if (a.flags & 2) {  // white_back: rgb_final + 1 - weights_sum
    const float extra = 1.f - wsum;
    F.x += extra; F.y += extra; F.z += extra; F.w += extra;
  }
  if (active) {
    *reinterpret_cast<float4*>(a.fea + ray * 32 + 4 * sub) = F;
    if (sub == 0 && a.depth) a.depth[ray] = depth;
  }
}

__global__ __launch_bounds__(256) void composite_bwd_kernel(CompArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int S = a.S, E = a.E;
  const int rays_per_block = blockDim.x / SEG;
  const int lr = threadIdx.x / SEG, sub = threadIdx.x % SEG;
  const long long ray_raw = (long long)blockIdx.x * rays_per_block + lr;
  const bool active = ray_raw < a.R;
  const long long ray = active ? ray_raw : a.R - 1;
  // per ray: z[E], x[E] (sigma + noise, sorted order), T[E], alpha[E], s[E], ord[E]
  float* zall = sm + lr * (6 * E);
  float* xs = zall + E;
  float* Ts = xs + E;
  float* al = Ts + E;
  float* ss = al + E;
  int* ord = reinterpret_cast<int*>(ss + E);

  for (int k = sub; k < E; k += SEG) {
    const int i = a.order_in ? a.order_in[ray * E + k] : k;     // no order given (no fine set): identity
    ord[k] = i;
    float zz, sg;
    if (a.feat_f) {
      zz = (i < S) ? a.z_f[ray * S + i] : a.z_c[ray * S + (i - S)];
      sg = (i < S) ? a.sig_f[ray * S + i] : a.sig_c[ray * S + (i - S)];
    } else {
      zz = a.z_c[ray * S + i];
      sg = a.sig_c[ray * S + i];
    }
    if (a.noise) sg += a.noise[ray * E + k] * a.noise_std;
    zall[k] = zz;   // sorted order
    xs[k] = sg;
  }
  __syncthreads();

  const float4 G = *reinterpret_cast<const float4*>(a.dfea + ray * 32 + 4 * sub);
  // white_back adds (1 - sum_k w_k) to every channel, last_back adds (1 - sum_k w_k) f_last (pigan_utils.py:261-268,
  // weights_sum taken before the last weight is topped up): dL/dw_k = G.f_k - [last_back] G.f_last - [white_back] sum(G)
  float gsum = G.x + G.y + G.z + G.w;
  gsum += __shfl_xor(gsum, 1);
  gsum += __shfl_xor(gsum, 2);
  gsum += __shfl_xor(gsum, 4);
  float wsum = 0.f;
  double T = 1.0;
  for (int k = 0; k < E; ++k) {
    const float delta = (k + 1 < E) ? (zall[k + 1] - zall[k]) : 1e10f;
    float dens = clamp_density(xs[k], a.clamp_mode);
    if (a.clamp_pin && a.clamp_mode == 0) dens = a.clamp_pin[ray * E + k] ? xs[k] : 0.f;
    const float alpha = 1.f - expf(-delta * dens);
    const float Tf = (float)T;
    wsum += alpha * Tf;
    T *= (double)(1.f - alpha + 1e-10f);
    const float4 f = *reinterpret_cast<const float4*>(feat_row(a, ray, ord[k]) + 4 * sub);
    float s = G.x * f.x + G.y * f.y + G.z * f.z + G.w * f.w;
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    s += __shfl_xor(s, 4);
    if (sub == 0) { Ts[k] = Tf; al[k] = alpha; ss[k] = s; }
  }
  __syncthreads();
  float Q = 0.f;
  const float s_off = ((a.flags & 1) ? ss[E - 1] : 0.f) + ((a.flags & 2) ? gsum : 0.f);
  for (int k = E - 1; k >= 0; --k) {
    const int i = ord[k];
    const float alpha = al[k], Tk = Ts[k], s = ss[k] - s_off;
    float w = alpha * Tk;
    if ((a.flags & 1) && k == E - 1) w += 1.f - wsum;       // the feature gradient of the last sample sees the topped-up weight

    const float dalpha = Tk * (s - Q);
    Q = fmaf(alpha, s, (1.f - alpha + 1e-10f) * Q);
    float* drow;
    float* dsg;
    if (a.feat_f) {
      drow = (i < S) ? a.dfeat_f + (ray * S + i) * 32 : a.dfeat_c + (ray * S + (i - S)) * 32;
      dsg = (i < S) ? a.dsig_f + ray * S + i : a.dsig_c + ray * S + (i - S);
    } else {
      drow = a.dfeat_c + (ray * S + i) * 32;
      dsg = a.dsig_c + ray * S + i;
    }
    if (active) *reinterpret_cast<float4*>(drow + 4 * sub) = make_float4(w * G.x, w * G.y, w * G.z, w * G.w);
    if ((k % SEG) == sub && active) {
      const float delta = (k + 1 < E) ? (zall[k + 1] - zall[k]) : 1e10f;
      const float x = xs[k];
      float dens = clamp_density(x, a.clamp_mode);
      float dgrad = clamp_density_grad(x, a.clamp_mode);
      if (a.clamp_pin && a.clamp_mode == 0) {
        const bool pass = a.clamp_pin[ray * E + k] != 0;
        dens = pass ? x : 0.f;
        dgrad = pass ? 1.f : 0.f;
      }
      // d alpha / d dens = delta * exp(-delta*dens)
      *dsg = dalpha * (delta * expf(-delta * dens)) * dgrad;
    }
  }
}

inline int rays_per_block_for(size_t bytes_per_ray) {
  int r = 32;
  while (r > 1 && (size_t)r * bytes_per_ray > 60 * 1024) r >>= 1;
  return r;
}

}  // namespace

extern "C" int cips_rays_fwd(const float* xg, const float* yg, const float* zg, float zc,
                             const float* cam2world, const float* jitter, float* points, float* z,
                             float* dirs, int B, int H, int W, int S, cips_stream_t stream) {
  if (B <= 0 || H <= 0 || W <= 0 || S <= 1) return (int)hipErrorInvalidValue;
  long long total = (long long)B * H * W * S;
  int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(rays_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, xg, yg, zg, zc,
                     cam2world, jitter, points, z, dirs, B, H, W, S);
  return CIPS_CHECK_LAUNCH();
}

extern "C" int cips_resample_fwd(const float* sigma, const float* z, const float* noise, float noise_std,
                                 const float* u, const float* origins, const float* dirs, float* fine_z,
                                 float* fine_pts, float* weights_out, float* cdf_out, long long* inds_out,
                                 int B, int n, int S, int clamp_mode, const float* cdf_in, const cips_ray_params* rays,
                                 cips_stream_t stream) {
  if (B <= 0 || n <= 0 || S < 3 || !fine_z || (!rays && (!origins || !dirs || !fine_pts))) return (int)hipErrorInvalidValue;
  ResampleArgs a;
  a.cdf_in = cdf_in;
  a.rg = RayGen{}; a.have_rg = 0;
  if (rays) { const int rc = fill_raygen(a.rg, rays); if (rc) return rc; if (a.rg.n != n) return (int)hipErrorInvalidValue; a.have_rg = 1; }
  a.sigma = sigma; a.z = z; a.noise = noise; a.noise_std = noise_std; a.u = u; a.origins = origins;
  a.dirs = dirs; a.fine_z = fine_z; a.fine_pts = fine_pts; a.weights_out = weights_out;
  a.cdf_out = cdf_out; a.inds_out = inds_out; a.B = B; a.n = n; a.S = S; a.clamp_mode = clamp_mode;
  size_t per_ray = (size_t)4 * S * sizeof(float);
  int rpb = rays_per_block_for(per_ray);
  long long R = (long long)B * n;
  int blocks = (int)((R + rpb - 1) / rpb);
  hipLaunchKernelGGL(resample_kernel, dim3(blocks), dim3(rpb * SEG), rpb * per_ray, (hipStream_t)stream, a);
  return CIPS_CHECK_LAUNCH();
}

extern "C" int cips_composite_fwd(const float* feat_c, const float* sig_c, const float* z_c,
                                  const float* feat_f, const float* sig_f, const float* z_f,
                                  const float* noise, float noise_std, float* fea, float* depth,
                                  float* weights, int* order, float* zsorted, int R, int S,
                                  int clamp_mode, int flags, const unsigned char* clamp_in, unsigned char* clamp_out,
                                  cips_stream_t stream) {
  if (R <= 0 || S <= 0) return (int)hipErrorInvalidValue;
  CompArgs a = {};
  a.feat_c = feat_c; a.sig_c = sig_c; a.z_c = z_c; a.feat_f = feat_f; a.sig_f = sig_f; a.z_f = z_f;
  a.noise = noise; a.noise_std = noise_std; a.fea = fea; a.depth = depth; a.weights = weights;
  a.order = order; a.zsorted = zsorted; a.R = R; a.S = S; a.E = feat_f ? 2 * S : S;
  a.clamp_mode = clamp_mode; a.flags = flags;
  a.clamp_pin = clamp_in; a.clamp_rec = clamp_out;
  size_t per_ray = (size_t)4 * a.E * sizeof(float);
  int rpb = rays_per_block_for(per_ray);
  int blocks = (R + rpb - 1) / rpb;
  hipLaunchKernelGGL(composite_fwd_kernel, dim3(blocks), dim3(rpb * SEG), rpb * per_ray, (hipStream_t)stream, a);
  return CIPS_CHECK_LAUNCH();
}

extern "C" int cips_composite_bwd(const float* feat_c, const float* sig_c, const float* z_c,
                                  const float* feat_f, const float* sig_f, const float* z_f,
                                  const float* noise, float noise_std, const int* order, const float* dfea,
                                  float* dfeat_c, float* dsig_c, float* dfeat_f, float* dsig_f, int R, int S,
                                  int clamp_mode, int flags, const unsigned char* clamp_in, cips_stream_t stream) {
  if (R <= 0 || S <= 0 || (!order && feat_f)) return (int)hipErrorInvalidValue;
  CompArgs a = {};
  a.feat_c = feat_c; a.sig_c = sig_c; a.z_c = z_c; a.feat_f = feat_f; a.sig_f = sig_f; a.z_f = z_f;
  a.noise = noise; a.noise_std = noise_std; a.order_in = order; a.dfea = dfea;
  a.dfeat_c = dfeat_c; a.dsig_c = dsig_c; a.dfeat_f = dfeat_f; a.dsig_f = dsig_f;
  a.R = R; a.S = S; a.E = feat_f ? 2 * S : S; a.clamp_mode = clamp_mode; a.flags = flags;
  a.clamp_pin = clamp_in; a.clamp_rec = nullptr;
  size_t per_ray = (size_t)6 * a.E * sizeof(float);
  int rpb = rays_per_block_for(per_ray);
  int blocks = (R + rpb - 1) / rpb;
  hipLaunchKernelGGL(composite_bwd_kernel, dim3(blocks), dim3(rpb * SEG), rpb * per_ray, (hipStream_t)stream, a);
  return CIPS_CHECK_LAUNCH();
}
