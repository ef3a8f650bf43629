// raygen.h — in-kernel ray set-up shared by the SIREN kernels (siren_bwd_x3.hip) and the resampler (render.hip).
#pragma once
#include "common.h"
#include "../../include/cips3d_hip.h"

// In-kernel ray set-up (what rays_kernel of render.hip materialises; exp/comm/comm_utils.py:365-438, 584-679): the
// sample point of (image b, ray, sample s) from the three linspace grids, the camera matrix and the jitter draw — 4 B
// per point read instead of 12 B, and no (B, n, S, 3) tensor in HBM.  Same expressions, in the same order, as
// rays_kernel.
struct RayGen {
  const float *xg, *yg, *zg;   // torch.linspace grids (W), (H), (S)
  const float* c2w;            // (B, 4, 4)
  const float* jitter;         // (B, n, S) uniforms or NULL
  const float* zvals;          // (B, n, S) depths or NULL; if set: point = camera origin + world ray direction * zvals[p]
                               // (the resampled "fine" points, generator_nerf_inr.py:590-592), grids / jitter unused
  float zc;
  int W, n, S;
};
struct RayDir { float dx, dy, dz; };
__device__ __forceinline__ RayDir ray_dir(const RayGen& g, int ray) {
  const int row = ray / g.W, col = ray - row * g.W;
  const float x = g.xg[col], y = g.yg[row];
  const float nrm = sqrtf(x * x + y * y + g.zc * g.zc);
  RayDir d = {x / nrm, y / nrm, g.zc / nrm};
  return d;
}
// camera-space sample at depth grid value z0 with jitter draw u (raw uniform; ignored when has_jit is false) -> world
// point and the jittered depth
__device__ __forceinline__ void ray_point(const RayGen& g, const float* M, const RayDir& d, float z0, float u, bool has_jit,
                                          float& wx, float& wy, float& wz, float& zout) {
  float z = z0;
  float px = d.dx * z, py = d.dy * z, pz = d.dz * z;
  if (has_jit) {
    const float off = (u - 0.5f) * (g.zg[1] - g.zg[0]);
    z = z + off;
    px = px + off * d.dx; py = py + off * d.dy; pz = pz + off * d.dz;
  }
  wx = ((M[0] * px + M[1] * py) + M[2] * pz) + M[3];
  wy = ((M[4] * px + M[5] * py) + M[6] * pz) + M[7];
  wz = ((M[8] * px + M[9] * py) + M[10] * pz) + M[11];
  zout = z;
}
// point-major index p = ray * S + s of image b -> world point (and its depth)
__device__ __forceinline__ void gen_point(const RayGen& g, int b, int p, float& wx, float& wy, float& wz, float& z) {
  const int ray = p / g.S, s = p - ray * g.S;
  const RayDir d = ray_dir(g, ray);
  const float* M = g.c2w + (long long)b * 16;
  if (g.zvals) {
    // origin + direction * depth, with the direction rotated like rays_kernel and the sum like resample_kernel
    z = g.zvals[(long long)b * g.n * g.S + p];
    const float dx = (M[0] * d.dx + M[1] * d.dy) + M[2] * d.dz;
    const float dy = (M[4] * d.dx + M[5] * d.dy) + M[6] * d.dz;
    const float dz = (M[8] * d.dx + M[9] * d.dy) + M[10] * d.dz;
    wx = M[3] + dx * z; wy = M[7] + dy * z; wz = M[11] + dz * z;
    return;
  }
  const float u = g.jitter ? g.jitter[(long long)b * g.n * g.S + p] : 0.f;
  ray_point(g, M, d, g.zg[s], u, g.jitter != nullptr, wx, wy, wz, z);
}
__device__ __forceinline__ void gen_point(const RayGen& g, int b, int p, float& wx, float& wy, float& wz) {
  float z;
  gen_point(g, b, p, wx, wy, wz, z);
}

static inline int fill_raygen(RayGen& g, const cips_ray_params* r) {
  if (!r || !r->xg || !r->yg || !r->zg || !r->cam2world || r->W <= 0 || r->H <= 0 || r->S <= 1) return (int)hipErrorInvalidValue;
  g.xg = r->xg; g.yg = r->yg; g.zg = r->zg; g.c2w = r->cam2world; g.jitter = r->jitter; g.zvals = r->zvals; g.zc = r->zc;
  g.W = r->W; g.n = r->W * r->H; g.S = r->S;
  return 0;
}
