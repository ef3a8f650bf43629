/*
 * cips3d_hip.h — C-ABI of libcips3d_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the CIPS-3D generator / discriminator hot path
 * (SURVEY.md §8b).  Every entry point takes raw DEVICE pointers (fp32,
 * row-major contiguous unless a leading dimension is given), sizes, scalar
 * parameters and a hipStream_t (passed as void*), returns a hipError_t as int
 * (0 = success), allocates nothing, keeps no state and is thread-safe.
 * The Python host (cips3d_amd/ops.py) binds these through ctypes; the reference
 * binds its two CUDA ops through pybind11 (exp/comm/op/fused_bias_act.cpp:11-21,
 * exp/comm/op/upfirdn2d.cpp:12-23) and everything else through ATen.
 *
 * Each declaration cites the reference code it replaces (paths relative to the
 * reference repo root).
 */
#ifndef CIPS3D_HIP_H
#define CIPS3D_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef void* cips_stream_t; /* hipStream_t */

/* ------------------------------------------------------------------ */
/* library info                                                        */
/* ------------------------------------------------------------------ */
int cips_version(void);          /* ABI version, bumped on signature change */
const char* cips_arch(void);     /* "gfx950" */

/* ------------------------------------------------------------------ */
/* H1  ray set-up                                                      */
/* replaces exp/comm/comm_utils.py:365-412 (get_initial_rays_trig),    */
/*          :416-438 (perturb_points), :584-679 (transform_sampled_points, */
/*          the three bmm's).  Camera sampling / cam2world (O(batch))  */
/*          stay on the host.                                          */
/* ------------------------------------------------------------------ */
/* xg[W], yg[H], zg[S]: torch.linspace grids built by the host (bit-exact
 * with the reference).  zc = -1/tan(fov/2).  cam2world (B,4,4).  jitter U
 * (B,n,S) in [0,1) or NULL (no perturbation).
 * out: points (B,n,S,3) world space, z (B,n,S), dirs (B,n,3) world space. */
int cips_rays_fwd(const float* xg, const float* yg, const float* zg, float zc,
                  const float* cam2world, const float* jitter,
                  float* points, float* z, float* dirs,
                  int B, int H, int W, int S, cips_stream_t stream);

/* ------------------------------------------------------------------ */
/* H2  fused FiLM-SIREN NeRF MLP (3 -> 128 -> 128 -> {sigma, 64 -> 32}) */
/* replaces exp/cips3d/models/generator.py:260-317                     */
/*   (NeRFNetwork.forward_with_frequencies_phase_shifts),              */
/*   exp/comm/models/film_layer.py:78-107 (FiLMLayer.forward),         */
/*   exp/comm/models/nerf_network.py:39-45 (UniformBoxWarp).           */
/* ------------------------------------------------------------------ */
typedef struct cips_siren_weights {
  const float* w0;  /* (128,3)   siren.network.0.linear.weight */
  const float* b0;  /* (128)                                  */
  const float* w1;  /* (128,128) siren.network.1.linear.weight */
  const float* b1;  /* (128)                                  */
  const float* ws;  /* (1,128)   siren.final_layer.weight      */
  const float* bs;  /* (1)                                    */
  const float* wc;  /* (64,128)  siren.color_layer_sine.linear.weight */
  const float* bc;  /* (64)                                   */
  const float* wf;  /* (32,64)   siren.color_layer_linear.0.weight */
  const float* bf;  /* (32)                                   */
  /* per-image FiLM vectors: gain = 15*gain_fc(style)+30, bias = bias_fc(style) */
  const float* g0;  /* (B,128) */
  const float* p0;  /* (B,128) */
  const float* g1;  /* (B,128) */
  const float* p1;  /* (B,128) */
  const float* gc;  /* (B,64)  */
  const float* pc;  /* (B,64)  */
  float box_scale;  /* 2/0.24 (UniformBoxWarp) */
  int trig_mode;    /* bit 0: 0 = Cody-Waite + minimax polynomial sin/cos, 1 = v_sin_f32 / v_cos_f32 (what the Python layer passes);
                       bit 1 (x3 forward kernels only): bf16 operand planes (2^-17) instead of the default fp16 planes (2^-22) —
                       kept for the A/B of tests/test_gpu_kernels.py::test_siren_forward_x3_sigma_is_fp32_class */
} cips_siren_weights;

/* points (B,P,3) -> feat (B,P,32), sigma (B,P). */
int cips_siren_fwd(const cips_siren_weights* w, const float* points,
                   float* feat, float* sigma, int B, int P, cips_stream_t stream);

/* Backward, stage 1 ("data" pass; recomputes the forward in-kernel, saves no
 * forward activations).  In: upstream grads dfeat (B,P,32), dsigma (B,P).
 * Out (HBM staging for the weight-gradient GEMMs; split-bf16 planes x = hi + lo, [B*P][F] row-major,
 *      i.e. the k-major operands of cips_gemm_bf16x3_km with K = points):
 *   h1, h2 (B*P,128), hc (B*P,64)         recomputed activations
 *   da2 (B*P,128), dac (B*P,64)           d loss / d sine-argument of layer 1 / colour layer
 * Out (per-partial-row reductions, partial rows = cips_siren_bwd_rows(B,P),
 *      each row belongs to one image: image = row / (rows/B)):
 *   red (rows, 868): [0,128) sum_p da1 | [128,512) sum_p da1*x_c (c=0..2, 128 each)
 *                    | [512,640) sum_p da2 | [640,704) sum_p dac | [704,832) sum_p dsigma*h2
 *                    | [832,864) sum_p dfeat | [864] sum_p dsigma
 */
int cips_siren_bwd_rows(int B, int P);
int cips_siren_bwd_data(const cips_siren_weights* w, const float* points,
                        const float* dfeat, const float* dsigma,
                        void* h1_hi, void* h1_lo, void* h2_hi, void* h2_lo, void* hc_hi, void* hc_lo,
                        void* da2_hi, void* da2_lo, void* dac_hi, void* dac_lo,
                        float* red, int B, int P, cips_stream_t stream);
/* The same data pass with the five staged tensors written as fp32 rows — h1, h2, da2: (B*P, 128); hc, dac: (B*P, 64) — for
 * weight-gradient contractions on the exact-fp32 MFMA GEMM (cips_gemm_f32, a_kmajor): the all-fp32 leg, in which no split
 * operand takes part (CIPS_SIREN_BWD=staged_f32 with CIPS_INR_MODE=f32 CIPS_SIREN_FWD=f32). */
int cips_siren_bwd_data_f32(const cips_siren_weights* w, const float* points, const float* dfeat, const float* dsigma, float* h1,
                            float* h2, float* hc, float* da2, float* dac, float* red, int B, int P, cips_stream_t stream);

/* Forward on the split-bf16 matrix-core chain (the default of the Python layer; same contract as cips_siren_fwd;
 * ~1e-5 relative instead of ~1e-6, 2.3x faster): */
int cips_siren_fwd_x3(const cips_siren_weights* w, const float* points, float* feat, float* sigma,
                      int B, int P, cips_stream_t stream);

/* Backward, fused bf16x3 form (default): forward recompute, data gradients, all three weight-gradient
 * contractions and every per-feature sum over points in one kernel on v_mfma_f32_32x32x16_bf16 with 3-pass
 * split operands (fp32 accumulate); no activation staging in HBM.
 * chunks = cips_siren_bwd_x3_chunks(B,P) workgroups per image; partials of image b are rows
 * [b*chunks, (b+1)*chunks) of both outputs:
 *   sred  (B*chunks, cips_siren_bwd_x3_sred())  floats [wave w=0..3][row 0..31][col 0..7], then 4 per-wave
 *         sums of dsigma (+4 pad).  Row r of wave w is feature 32w + r; columns:
 *           0 sum_p da1 | 1..3 sum_p da1 * (x, y, z) | 4 sum_p da2 | 5 sum_p dsigma * h2
 *           | 6 sum_p dac (waves 0,1: 64 features) | 7 sum_p dfeat (waves 0 and 2: partial sums, 32 channels)
 *   gpart (B*chunks, cips_siren_bwd_x3_gpart()) floats:
 *         [0,16384)  da2^T h1 (128,128) | [16384,24576) dac^T h2 (64,128)
 *         | [24576,26624) and [26624,28672) two partial dfeat^T hc (32,64)
 * (da1, da2, dac = d loss / d sine argument of layer 0, layer 1, colour layer.)
 */
int cips_siren_bwd_x3_chunks(int B, int P);
int cips_siren_bwd_x3_gpart(void);
int cips_siren_bwd_x3_sred(void);
int cips_siren_bwd_x3(const cips_siren_weights* w, const float* points, const float* dfeat,
                      const float* dsigma, float* sred, float* gpart, int B, int P, cips_stream_t stream);

/* The 16 gradient tensors of the SIREN from the fused backward's per-workgroup partials, in one launch (chain rule of
 * exp/comm/models/film_layer.py:88-107): per-image FiLM gradients dg* / dp* (B,128 | B,64) and the batch-summed weight /
 * bias gradients in the parameters' own shapes. */
typedef struct cips_siren_grads {
  float *dg0, *dp0, *dg1, *dp1, *dgc, *dpc;          /* (B,128) x4, (B,64) x2 */
  float *dw0, *db0, *dw1, *db1, *dws, *dbs, *dwc, *dbc, *dwf, *dbf;   /* (128,3) (128) (128,128) (128) (1,128) (1) (64,128) (64) (32,64) (32) */
} cips_siren_grads;
int cips_siren_bwd_x3_finalize(const cips_siren_weights* w, const float* sred, const float* gpart, int B, int chunks,
                               const cips_siren_grads* out, cips_stream_t stream);

/* Ray parameters for in-kernel point generation (what cips_rays_fwd materialises): the sample point of
 * (image b, ray r, sample s) is recomputed from the linspace grids, the camera matrix and the jitter draw, 4 B per
 * point read instead of 12 B and no (B, n, S, 3) tensor in HBM.  Point index p = r * S + s. */
typedef struct cips_ray_params {
  const float* xg; const float* yg; const float* zg;   /* torch.linspace grids (W), (H), (S) */
  const float* cam2world;                                /* (B,4,4) */
  const float* jitter;                                   /* (B,n,S) uniforms in [0,1) or NULL */
  const float* zvals;                                    /* (B,n,S) depths or NULL; if set: point = camera origin + world ray
                                                            direction * zvals[p] (the resampled fine points,
                                                            exp/dev/nerf_inr/models/generator_nerf_inr.py:590-592) */
  float zc;                                              /* -1/tan(fov/2) */
  int H, W, S;
} cips_ray_params;

/* cips_siren_fwd_x3 with the points generated in-kernel from `rays`; zout (optional, (B,P)) receives their depths. */
int cips_siren_fwd_x3_rays(const cips_siren_weights* w, const cips_ray_params* rays, float* feat, float* sigma,
                           float* zout, int B, cips_stream_t stream);

/* cips_siren_bwd_x3 with the points generated in-kernel from `rays` (P = H*W*S points per image). */
int cips_siren_bwd_x3_rays(const cips_siren_weights* w, const cips_ray_params* rays, const float* dfeat,
                           const float* dsigma, float* sred, float* gpart, int B, cips_stream_t stream);

/* Fused ray-march for NON-hierarchical sampling: ray set-up + FiLM-SIREN + alpha-composite in one kernel that walks
 * the samples along the ray (a wave owns 32 rays, one lane pair per ray; the running transmittance / feature / depth
 * accumulators live in registers).  Replaces, for hierarchical_sample=False,
 *   exp/comm/comm_utils.py:365-438, 584-679 (rays), exp/cips3d/models/generator.py:260-317 (SIREN),
 *   exp/pigan/pigan_utils.py:212-273 (fancy_integration; the merge of generator.py:1733-1752 is the identity).
 * noise (B,n,S) standard normals or NULL; clamp_mode 0 relu / 1 softplus; flags bit0 last_back, bit1 white_back.
 * out: fea (B,n,32), depth (B,n) [may be NULL]; optional: weights (B,n,S), and — for a training forward whose backward
 * needs them — the per-sample feat (B,P,32), sigma (B,P), z (B,P).  With those NULL the kernel moves 4*S + 132 B per ray
 * (SURVEY.md §8d-iii).  clamp_in / clamp_out: optional (B*n, S) branch masks of the relu clamp, as in cips_composite_fwd
 * (a separate instantiation of the kernel: the production launch, both NULL, keeps its registers). */
int cips_march_fwd_x3(const cips_siren_weights* w, const cips_ray_params* rays, const float* noise, float noise_std,
                      int clamp_mode, int flags, float* fea, float* depth, float* weights, float* feat, float* sigma,
                      float* z, int B, const unsigned char* clamp_in, unsigned char* clamp_out, cips_stream_t stream);

/* ------------------------------------------------------------------ */
/* H3  hierarchical resampling + merge + alpha-composite               */
/* ------------------------------------------------------------------ */
/* Coarse weights + inverse-CDF resampling.
 * replaces exp/dev/nerf_inr/models/generator_nerf_inr.py:537-598
 *   (get_fine_points_and_direction), exp/pigan/pigan_utils.py:164-209
 *   (sample_pdf) and the weights part of :212-273 (fancy_integration).
 * sigma (R,S), z (R,S), noise (R,S) or NULL (noise already scaled by
 * noise_std is NOT assumed: kernel applies sigma + noise*noise_std),
 * u (R,S) uniforms, origins (B,3), dirs (R,3) with R = B*n rays.
 * out: fine_z (R,S), fine_pts (R,S,3); optional debug outs (may be NULL):
 * weights (R,S), cdf (R,S-1), inds (R,S) int64 (searchsorted result).
 * clamp_mode: 0 = relu, 1 = softplus.
 * cdf_in (optional, (R,S-1)): take the cdf from the caller instead of computing it — the "bit-exact integer bookkeeping
 * on identical float inputs" contract: with the reference's cdf and u the indices must equal
 * torch.searchsorted's exactly.
 * rays (optional): ray directions and origins are recomputed from the ray parameters (origins / dirs may then be NULL,
 * and fine_pts may be NULL when the fine pass regenerates its points from fine_z, cips_ray_params.zvals). */
int cips_resample_fwd(const float* sigma, const float* z, const float* noise, float noise_std,
                      const float* u, const float* origins, const float* dirs,
                      float* fine_z, float* fine_pts,
                      float* weights_out, float* cdf_out, long long* inds_out,
                      int B, int n, int S, int clamp_mode, const float* cdf_in, const cips_ray_params* rays,
                      cips_stream_t stream);

/* Merge (coarse + fine, ascending z) and alpha-composite.
 * replaces exp/cips3d/models/generator.py:1733-1752 (cat/sort/gather) and
 *   exp/pigan/pigan_utils.py:212-273 (fancy_integration).
 * feat_c (R,S,32), sig_c (R,S), z_c (R,S); fine set same shapes or NULL
 * (then E = S, no merge).  noise (R,E) in SORTED order or NULL.
 * out: fea (R,32), depth (R), weights (R,E) sorted order, order (R,E) int32
 * (index into [fine(0..S-1), coarse(S..2S-1)] like torch.cat([fine, coarse]);
 * for the non-hierarchical case the identity), zsorted (R,E) (may be NULL).
 * flags: bit0 last_back, bit1 white_back.
 * clamp_in / clamp_out (optional, (R,E) bytes, relu clamp only; no reference counterpart — pigan_utils.py:246-252 is the
 * clamp they describe): relu(sigma + nerf_noise * eps) is a discontinuity of the gradient — two evaluations whose
 * pre-activations differ by rounding may take different branches.  With clamp_in the branch of sample (ray, sorted
 * position k) is clamp_in[ray*E + k] (0 = clamped, else the linear branch) instead of the sign of the value computed here;
 * clamp_out receives the branch taken.  Both NULL in production; the parity tests pin the CPU oracle's branches through
 * them.  They are arguments of the call (round 5; rounds 3-4 had a process-global hook): the library keeps no state. */
int cips_composite_fwd(const float* feat_c, const float* sig_c, const float* z_c,
                       const float* feat_f, const float* sig_f, const float* z_f,
                       const float* noise, float noise_std,
                       float* fea, float* depth, float* weights, int* order, float* zsorted,
                       int R, int S, int clamp_mode, int flags, const unsigned char* clamp_in, unsigned char* clamp_out,
                       cips_stream_t stream);

/* Backward of the above w.r.t. feat/sigma of both sample sets (z has no grad:
 * fine z is detach()ed, generator_nerf_inr.py:575-579; coarse z is an input).
 * dfea (R,32) upstream.  Re-reads the forward inputs (no saved activations
 * besides `order`; order == NULL with no fine set means the identity).  clamp_in as in cips_composite_fwd. */
int cips_composite_bwd(const float* feat_c, const float* sig_c, const float* z_c,
                       const float* feat_f, const float* sig_f, const float* z_f,
                       const float* noise, float noise_std, const int* order,
                       const float* dfea,
                       float* dfeat_c, float* dsig_c, float* dfeat_f, float* dsig_f,
                       int R, int S, int clamp_mode, int flags, const unsigned char* clamp_in, cips_stream_t stream);

/* ------------------------------------------------------------------ */
/* generic batched fp32 GEMM on v_mfma_f32_32x32x2_f32 with fused epilogues */
/* the workhorse under H4 (bmm in exp/comm/models/mod_conv_fc.py:489),  */
/* the SIREN weight gradients, and H5 convolutions (im2col GEMM).       */
/* ------------------------------------------------------------------ */
typedef struct cips_gemm_desc {
  /* C[b] (M,N) = epilogue( A[b] (M,K) @ B[b] (K,N) ) */
  const float* A; const float* B; float* C;
  int M, N, K;
  int lda, ldb, ldc;
  long long strideA, strideB, strideC; /* elements between batches */
  int batch;
  int a_kmajor;          /* 0: A stored (M,K) row-major; 1: A stored (K,M) row-major ("TN") */
  int b_nmajor;          /* 0: B stored (K,N) row-major; 1: B stored (N,K) row-major ("NT") */
  /* ---- epilogue (all optional; every aux matrix uses ldc / strideC addressing) ---- */
  float alpha;           /* acc *= alpha (0 is treated as 1) */
  const float* bias;     /* (N) added after alpha */
  const float* bias_m;   /* (M) per-row bias (conv output channel when C is (O, Ho*Wo)) */
  int act;               /* 0 none; 1 leaky_relu(slope) ; 2 leaky_relu(slope)*act_gain */
  float slope; float act_gain;
  const float* resid;    /* after act: C2 = act(..) + resid */
  float* C2;             /* second output (written only if non-NULL) */
  const float* add;      /* before mask: s = acc + add */
  const float* rgb_g;    /* (batch*M,3) with row stride 3: s += rgb_g[m,:] @ rgb_w[:, n] */
  const float* rgb_w;    /* (3,N) row-major */
  float* C_unmasked;     /* if non-NULL, s stored here before masking */
  const float* mask;     /* C = s * (mask > 0 ? 1 : slope) * (act==2? act_gain:1) */
} cips_gemm_desc;

int cips_gemm_f32(const cips_gemm_desc* d, cips_stream_t stream);

/* ------------------------------------------------------------------ */
/* fp32-grade GEMM on the bf16 matrix cores by 3-pass operand splitting */
/* ("bf16x3": x = hi + lo in two bf16 planes; a*b ~ ah*bh + ah*bl + al*bh, */
/* fp32 accumulate; ~1e-5 relative per layer against the 1e-3 parity bar). */
/* Used for the 17 modulated 512x512 layers of the CIPS INR head        */
/* (torch.bmm in exp/comm/models/mod_conv_fc.py:489 and its backward).  */
/* ------------------------------------------------------------------ */
typedef struct cips_gemm_x3_desc {
  /* C[b][m][n] = epilogue( sum_k A[b][m][k] * B[b][n][k] ): both operands contraction-contiguous ("NT"),
   * each given as two bf16 planes (uint16 storage). */
  const void* A_hi; const void* A_lo; const void* B_hi; const void* B_lo;
  int M, N, K;                 /* K % 32 == 0 */
  int lda, ldb;                /* elements; multiples of 8 */
  long long strideA, strideB;  /* elements between batches; multiples of 8 */
  int batch;
  /* outputs, any may be NULL */
  float* C; int ldc; long long strideC;                /* fp32 row-major */
  void* P_hi; void* P_lo; int ldp; long long strideP;  /* split planes, row-major [M][ldp] */
  void* T_hi; void* T_lo; int ldt; long long strideT;  /* split planes, transposed [N][ldt] */
  void* mask_out;              /* bf16 plane [M][ldp] of the value right after `act` (LeakyReLU gate source) */
  /* epilogue, in this order: +add, +rgb term, store C_unmasked, *gate(mask), act, store mask_out, +res, outputs */
  const float* add;            /* fp32 [M][ldc] */
  const float* rgb_g; const float* rgb_w;   /* (batch*M,3), (3,N) */
  float* C_unmasked;           /* fp32 [M][ldc] */
  const void* mask;            /* bf16 plane [M][ldp]: v *= (mask > 0 ? 1 : slope) */
  int act; float slope;        /* act 1: leaky_relu(slope) */
  const void* res_hi; const void* res_lo;   /* residual planes [M][ldp], added after act */
  /* gate planes as BIT planes (1 bit per element, bit c&7 of byte [m][c>>3], rows of ldp/8 bytes, batch stride
   * strideP/8): bit 0 of gate_bits = `mask` is one, bit 1 = `mask_out` is written as one (value > 0).  Needs
   * N % 32 == 0, ldp % 32 == 0, strideP % 32 == 0.  0 = bf16 planes as above. */
  int gate_bits;
  /* ToRGB forward folded into the epilogue (ABI 3; generator.py:949-1006, 1139-1144: rgb += out . T^T + tau): with
   * torgb_w (3, N) set, every 128-column block j of the FINAL value of a row (what P receives) leaves the partial
   * products  torgb_part[j][b*M + m][c] = sum_{n in block j} value[m][n] * torgb_w[c][n]  (c = 0..2, 4 floats per row,
   * the 4th is 0); torgb_part holds (N/128) * batch*M * 4 floats.  cips_torgb_finish adds the blocks in order, the
   * bias and (optionally) the running rgb.  Only the 256x256-tile v3 kernel implements it: cips_gemm_bf16x3 returns
   * hipErrorNotSupported for any other shape (use cips_torgb_fwd_x3 on the planes then). */
  const float* torgb_w; float* torgb_part;
  /* The addend as the split planes of a GATED tensor (ABI 4): the skip gradient of the head's backward is the previous
   * layer's un-gated gradient D, of which that layer already wrote the gated planes P = D * (gate ? 1 : slope) for its
   * own GEMMs.  With addp_hi / addp_lo ([M][ldp] planes, batch stride strideP) and addp_gate (the bit plane of that
   * gate, same layout as `mask` with gate_bits bit 0) set, the epilogue adds  (hi + lo) * (bit ? 1 : addp_gain)
   * (addp_gain = 1 / slope) in place of `add` — D is recovered to the planes' 2^-17 relative precision and the
   * producer need not write, nor this kernel read, a separate fp32 copy (C_unmasked / add: 2 x 4 bytes per element
   * saved).  Exclusive with `add`; needs `mask` as a bit plane; v3 kernel only (hipErrorNotSupported elsewhere). */
  const void* addp_hi; const void* addp_lo; const void* addp_gate; float addp_gain;
  /* Kernel choice (ABI 5; replaces the process-global cips_gemm_bf16x3_set_wide and the CIPS_X3_* environment variables):
   * 0 = automatic (production): 256x256 tiles — the v3 schedule for interior shapes, else the ragged-shape wide kernel —
   * for problems that fill the chip with them, the 256x128 kernel otherwise; 1 = the 256x128 kernel only; 2 = 256x256 tiles
   * whenever a kernel takes the shape; 3 = like 2 but never the v3 schedule.  1-3 serve the parity tests of each kernel.
   * The K-major entry points read it the same way (1: never the 256x256-tile kernel, 2 / 3: whenever the shape allows). */
  int kernel;
} cips_gemm_x3_desc;

int cips_gemm_bf16x3(const cips_gemm_x3_desc* d, cips_stream_t stream);
/* 1 when cips_gemm_bf16x3 would run this descriptor on the v3 kernel (the only one with the fused ToRGB partials) */
int cips_gemm_bf16x3_fuses_torgb(const cips_gemm_x3_desc* d);
/* 1 when cips_gemm_bf16x3 would take this descriptor's planes addend (addp_*; v3 kernel only) */
int cips_gemm_bf16x3_takes_addp(const cips_gemm_x3_desc* d);
/* rgb[m][c] = (accumulate ? rgb[m][c] : 0) + bias[c] + sum_j part[j][m][c];  part (nblocks, M, 4), rgb (M, 3) */
int cips_torgb_finish(const float* part, int nblocks, const float* bias, float* rgb, long long M, int accumulate,
                      cips_stream_t stream);
/* K-major form: C[b][m][n] = sum_k A[b][k][m] * B[b][k][n] (A planes [K][lda], B planes [K][ldb]: the row-major
 * activation / gradient planes themselves; LDS transpose reads build the fragments).  fp32 C output only. */
int cips_gemm_bf16x3_km(const cips_gemm_x3_desc* d, cips_stream_t stream);
/* Up to four K-major problems of identical shape (M, N, K, batch, leading dimensions, strides) in one launch of
 * 256x256 tiles; only the operand planes and C differ.  hipErrorNotSupported (801) when the shapes do not qualify:
 * issue them one by one with cips_gemm_bf16x3_km then. */
int cips_gemm_bf16x3_km_grouped(const cips_gemm_x3_desc* descs, int ngroups, cips_stream_t stream);

/* Implicit-GEMM convolution on the split-bf16 NT kernel (EqualConv2d forward, exp/cips3d/models/discriminator.py:40-48,
 * and — with the flipped, transposed weights — its data gradient at stride 1):
 *   y[b][o][oy*Wo+ox] = sum_{ky,kx,c} w[o][(ky*kw+kx)*C + c] * x[b][oy*stride-pad+ky][ox*stride-pad+kx][c]
 * x: NHWC split planes of B*H*W + 1 rows of C (the extra LAST ROW MUST BE ZERO: it stands in for the padding),
 * w: planes [O][kh*kw*C]; y: fp32 NCHW (B, O, Ho, Wo).  C % 32 == 0, Ho*Wo % 8 == 0; else hipErrorNotSupported. */
typedef struct cips_conv_x3_desc {
  const void* w_hi; const void* w_lo;
  const void* x_hi; const void* x_lo;
  float* y;
  int B, C, H, W, O, kh, kw, stride, pad;
  int ksplit;      /* <= 1: off.  > 1: the contraction (kh*kw*C) is cut into ksplit ranges computed by different workgroups */
  float* part;     /* (small output planes leave most CUs idle otherwise); part: ksplit * B*O*Ho*Wo floats of scratch, */
                   /* summed into y by the call.  cips_conv2d_x3_ksplit proposes a count for (B, O, N = Ho*Wo, K).   */
  const float* bias;  /* optional (O): added to every pixel of channel o                                                  */
  int act;            /* 0: none; 1: y = leaky_relu(y + bias, slope) * act_scale — EqualConv2d followed by FusedLeakyReLU */
  float slope, act_scale;   /* (discriminator.py:205-215, fused_act.py:47-86) in the GEMM epilogue                       */
} cips_conv_x3_desc;
int cips_conv2d_x3(const cips_conv_x3_desc* d, cips_stream_t stream);
int cips_conv2d_x3_ksplit(int B, int O, int N, int K);
/* Data gradient of a STRIDE-2, UNPADDED convolution (the EqualConv2d behind a Blur: exp/cips3d/models/discriminator.py:190-203)
 * without col2im: input pixel (P, Q) only receives the taps ky = P mod 2 (+2, ...), kx = Q mod 2 (+2, ...), so each of the four
 * parity classes (a, b) = (P mod 2, Q mod 2) is a small STRIDE-1 convolution over dy with its own filter bank:
 *   dxp_ab[i][c][U*Ws_b + V] = sum_{ty < Ta, tx < Tb, o} bank_ab[c][(ty*Tb + tx)*O + o] * dy[i][U - (Ta-1) + ty][V - (Tb-1) + tx][o]
 *                            = dx[i][c][2U + a][2V + b]
 * with Ta = ceil((kh-a)/2), Tb = ceil((kw-b)/2), Hs_a = ceil((H-a)/2), Ws_b = ceil((W-b)/2) and
 *   bank_ab[c][(ty*Tb + tx)*O + o] = w[o][c][a + 2(Ta-1-ty)][b + 2(Tb-1-tx)]   (planes, row pitch Ta*Tb*O, at element offset w_off[2a+b]).
 * All four run in ONE persistent launch of the implicit-GEMM kernel (longest contraction first).  dy: NHWC split planes
 * [B*Ho*Wo + 1][O] with a ZERO LAST ROW (Ho = (H-kh)/2 + 1).  Output: four compact blocks, class (a, b) at element offset
 * out_off[2a+b] of dxp as fp32 (B, C, Np_ab), Np_ab = Hs_a*Ws_b rounded up to a multiple of 8 (the padding elements are written
 * as zeros); cips_upfirdn2d_parity reads them in place of the interleaved (B, C, H, W) tensor.  O % 32 == 0, C % 8 == 0. */
typedef struct cips_conv_dgrad_s2_desc {
  const void* w_hi; const void* w_lo;
  const void* dy_hi; const void* dy_lo;
  float* dxp;
  int B, C, H, W, O, kh, kw;         /* H, W: the convolution's INPUT size */
  long long w_off[4], out_off[4];
} cips_conv_dgrad_s2_desc;
int cips_conv2d_x3_dgrad_s2(const cips_conv_dgrad_s2_desc* d, cips_stream_t stream);
/* Operand planes of convolution WEIGHTS for cips_conv2d_x3 / cips_conv2d_x3_dgrad_s2, up to cips_conv_weight_prep_max_jobs() layers
 * per launch (EqualConv2d: `weight * scale`, exp/cips3d/models/discriminator.py:33, 44): w (O, C, kh, kw) fp32 ->
 *   fwd_hi / fwd_lo: planes [O][kh*kw*C], contraction index (ky, kx, c)                                   (NULL: not written)
 *   alt_kind 1: alt_hi / alt_lo = planes [C][kh*kw*O] of the flipped, channel-transposed bank: element (c, (kh-1-ky, kw-1-kx, o))
 *               — the stride-1 data gradient's filter bank;
 *   alt_kind 2: alt_hi / alt_lo = the four parity banks of cips_conv2d_x3_dgrad_s2 at element offsets bank_off[2a+b];
 *   alt_kind 0: no alternate form.
 * hi = bf16(w * scale) (round to nearest even), lo = bf16(w * scale - hi): the values of a separate multiply + cips_split_planes. */
#define CIPS_WPREP_MAX_JOBS 24
typedef struct cips_wprep_job {
  const float* w;
  void* fwd_hi; void* fwd_lo;
  void* alt_hi; void* alt_lo;
  float scale;
  int O, C, kh, kw, alt_kind;
  int bank_off[4];
} cips_wprep_job;
int cips_conv_weight_prep_max_jobs(void);
int cips_conv_weight_prep_batch(const cips_wprep_job* jobs, int njobs, cips_stream_t stream);
/* Weight gradient of the same convolution on the K-major kernel (contraction over all B*Ho*Wo output pixels, split in
 * `nchunks` ranges whose partial sums the caller adds):
 *   part[chunk][ky*kw+kx][o][c] = sum_{q in chunk} dy[q][o] * x[pixel(q)*stride - pad + (ky,kx)][c]
 * dy: NHWC split planes [B*Ho*Wo][O]; x: NHWC split planes [B*H*W + 1][C] with a ZERO LAST ROW; part: fp32
 * (nchunks, kh*kw, O, C).  The pixel range is cut at 32-row granularity into nchunks nearly equal ranges (chunk c takes
 * k-tiles [c*T/nchunks, (c+1)*T/nchunks), T = B*Ho*Wo/32).  O, C multiples of 8, B*Ho*Wo % 32 == 0, nchunks <= T; else
 * hipErrorNotSupported. */
typedef struct cips_conv_wgrad_desc {
  const void* dy_hi; const void* dy_lo;
  const void* x_hi; const void* x_lo;
  float* part;
  int B, C, H, W, O, kh, kw, stride, pad, nchunks;
} cips_conv_wgrad_desc;
int cips_conv2d_x3_wgrad(const cips_conv_wgrad_desc* d, cips_stream_t stream);
/* Tail of the weight gradient: dw (O, C, kh, kw) = scale * sum over chunks of part (nchunks, kh*kw, O, C); scale is
 * EqualConv2d's 1/sqrt(C k^2) (discriminator.py:33, 44), which the autograd of `weight * scale` applies to the gradient. */
int cips_conv_wgrad_finish(const float* part, float* dw, int nchunks, int taps, int O, int C, float scale,
                           cips_stream_t stream);
/* Activation operand of cips_conv2d_x3 / cips_conv2d_x3_wgrad: x (B, C, n = H*W) fp32 NCHW -> NHWC split planes
 * t_hi, t_lo (B*n + 1, C) bf16, the last row zero (read wherever a tap falls into the padding).  C % 8 == 0 (n % 4 == 0 takes
 * 16-byte loads, any other n scalar loads). */
int cips_split_planes_nhwc(const float* x, void* t_hi, void* t_lo, int B, int C, int n, cips_stream_t stream);

/* fp32 (rows, cols) [ldx] -> split bf16 planes row-major [rows][ldp] and/or transposed [cols][ldt]. */
int cips_split_planes(const float* x, void* p_hi, void* p_lo, void* t_hi, void* t_lo, int rows, int cols,
                      int ldx, int ldp, int ldt, int batch, long long stride_x, long long stride_p,
                      long long stride_t, cips_stream_t stream);

/* ------------------------------------------------------------------ */
/* H4  CIPS INR head helpers (modulated FC, demodulated)                */
/* replaces exp/comm/models/mod_conv_fc.py:470-489 (SinStyleMod.forward_bmm) */
/* ------------------------------------------------------------------ */
/* weight (in,out) [= SinStyleMod.weight[0]], style s (B,in) [= modulation(style)].
 * out: wb (B,in,out) = W*(s+1)*demod, wbt (B,out,in) its transpose,
 * demod (B,out) = rsqrt(sum_in (W*(s+1))^2 + eps). */
int cips_modfc_prep(const float* weight, const float* s, float* wb, float* wbt, float* demod,
                    int B, int in_dim, int out_dim, float eps, cips_stream_t stream);
/* same, writing the bf16x3 operand planes: wb [in][out] and wbt [out][in], hi/lo each. */
int cips_modfc_prep_x3(const float* weight, const float* s, void* wb_hi, void* wb_lo, void* wbt_hi,
                       void* wbt_lo, float* demod, int B, int in_dim, int out_dim, float eps,
                       cips_stream_t stream);
/* backward of prep: gwb (B,in,out) = dL/d wb  ->  dweight (in,out), ds (B,in).
 * cbuf: caller-provided scratch of B*out floats. */
int cips_modfc_prep_bwd(const float* weight, const float* s, const float* demod, const float* gwb,
                        float* cbuf, float* dweight, float* ds, int B, int in_dim, int out_dim,
                        cips_stream_t stream);

/* Batched forms: all modulated-FC layers of the CIPS head in one call (njobs <= cips_modfc_max_jobs()); same
 * per-layer contracts as cips_modfc_prep_x3 / cips_modfc_prep_bwd, one shared batch size B. */
typedef struct cips_modfc_prep_job {
  const float* weight; const float* s;        /* (in,out), (B,in) */
  void *wb_hi, *wb_lo, *wbt_hi, *wbt_lo;      /* (B,in,out) and (B,out,in) bf16 planes */
  float* demod;                               /* (B,out) */
  int in_dim, out_dim;
} cips_modfc_prep_job;
typedef struct cips_modfc_bwd_job {
  const float* weight; const float* s; const float* demod; const float* gwb;   /* gwb (B,in,out) = dL/d wb */
  float *cbuf, *dweight, *ds;                 /* scratch (B,out); outputs (in,out), (B,in) */
  int in_dim, out_dim;
} cips_modfc_bwd_job;
int cips_modfc_max_jobs(void);
int cips_modfc_prep_x3_batch(const cips_modfc_prep_job* jobs, int njobs, int B, float eps, cips_stream_t stream);
int cips_modfc_prep_bwd_batch(const cips_modfc_bwd_job* jobs, int njobs, int B, cips_stream_t stream);
/* Co-resident form of cips_modfc_prep_bwd_batch (same results up to the summation order): kernels without LDS and with
 * <= 40 VGPRs, which get wave slots BESIDE a kernel that owns the whole LDS (the fused SIREN backward) instead of waiting
 * for it — for a caller that runs this tail on a side stream.  cbuf must hold cips_cores_colsum_parts() planes of (B, out).
 * B <= 64, out_dim % 4 == 0, else hipErrorNotSupported. */
int cips_cores_colsum_parts(void);
int cips_modfc_prep_bwd_batch_cores(const cips_modfc_bwd_job* jobs, int njobs, int B, cips_stream_t stream);

/* Grouped small Linear layers: every per-image vector of the hot path is a Linear of a style vector — the 18
 * SinStyleMod.modulation layers of the CIPS head (exp/comm/models/mod_conv_fc.py:433-436, 474) and the gain_fc / bias_fc
 * pairs of the FiLM layers (exp/comm/models/film_layer.py:59-63, 88-93).  One launch for all of them:
 *   forward   y_j (B,out_j) = x_j (B,in_j) w_j^T (out_j,in_j) + bias_j            (bias may be NULL)
 *   backward  dw_j = dy_j^T x_j, db_j = sum_b dy_j (db may be NULL), and — when dx != NULL, for jobs that all share one
 *             x — dx (B,in) = sum_j dy_j w_j (partials in `scratch`, cips_grouped_linear_scratch() floats, summed in
 *             a fixed order).  in_j % 4 == 0, in_j <= 512, njobs <= cips_grouped_linear_max_jobs(). */
typedef struct cips_glin_job {
  const float* x; const float* w; const float* bias; float* y;      /* forward */
  const float* dy; float* dw; float* db;                             /* backward */
  int in_dim, out_dim;
} cips_glin_job;
int cips_grouped_linear_max_jobs(void);
int cips_grouped_linear_fwd(const cips_glin_job* jobs, int njobs, int B, cips_stream_t stream);
long long cips_grouped_linear_scratch(const cips_glin_job* jobs, int njobs, int B);
int cips_grouped_linear_bwd(const cips_glin_job* jobs, int njobs, int B, float* dx, float* scratch,
                            long long scratch_floats, cips_stream_t stream);

/* Row-wise normalisation + activation of the z -> style mapping MLPs (exp/cips3d/models/multi_head_mapping.py:13-19
 * PixelNorm; :62-84 LayerNorm / LeakyReLU(0.2) after every Linear).  x, y (rows, cols <= 1024); stats (rows, 2) scratch
 * kept for the backward.  mode bit 0: LayerNorm (eps 1e-5, affine gamma / beta), bit 1: LeakyReLU(slope) after it,
 * bit 2: PixelNorm y = x * rsqrt(mean(x^2) + 1e-8) (alone).  Backward: dx (and d gamma, d beta through the (rows, cols)
 * scratch dyhat when bit 0 is set); `y` is the forward's output (the LeakyReLU gate is its sign). */
int cips_rownorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stats, int rows, int cols,
                     int mode, float slope, cips_stream_t stream);
int cips_rownorm_bwd(const float* x, const float* y, const float* gamma, const float* stats, const float* dy, float* dx,
                     float* dyhat, float* dgamma, float* dbeta, int rows, int cols, int mode, float slope,
                     cips_stream_t stream);

/* ToRGB (generator.py:983-1006): rgb (M,3) (+)= x (M,K) @ w^T (3,K) + bias.
 * accumulate != 0: rgb += ...  */
int cips_torgb_fwd(const float* x, const float* w, const float* bias, float* rgb,
                   long long M, int K, int accumulate, cips_stream_t stream);
/* dw (3,K) = drgb^T @ x, dbias(3) = sum drgb.  Uses `partials` scratch of
 * cips_torgb_bwd_partials(M) * 4 * K floats, reduced deterministically. */
int cips_torgb_bwd_partials(long long M);
int cips_torgb_bwd_w(const float* x, const float* drgb, float* partials, float* dw, float* dbias,
                     long long M, int K, cips_stream_t stream);
/* split-plane input variants of the two ToRGB reductions (x = x_hi + x_lo, bf16 planes [M][K]). */
int cips_torgb_fwd_x3(const void* x_hi, const void* x_lo, const float* w, const float* bias, float* rgb,
                      long long M, int K, int accumulate, cips_stream_t stream);
int cips_torgb_bwd_w_x3(const void* x_hi, const void* x_lo, const float* drgb, float* partials, float* dw,
                        float* dbias, long long M, int K, cips_stream_t stream);
/* the taps of up to 8 blocks (same M; K = 512, else hipErrorNotSupported) against one drgb in two launches:
 * x_hi / x_lo: host arrays of njobs device plane pointers; partials (njobs, chunks, 4, K); dw (njobs, 3, K); dbias (njobs, 3) */
int cips_torgb_bwd_w_x3_batch(const void* const* x_hi, const void* const* x_lo, int njobs, const float* drgb,
                              float* partials, float* dw, float* dbias, long long M, int K, cips_stream_t stream);
/* co-resident form (see cips_modfc_prep_bwd_batch_cores): same arguments, layouts and scratch */
int cips_torgb_bwd_w_x3_batch_cores(const void* const* x_hi, const void* const* x_lo, int njobs, const float* drgb,
                                    float* partials, float* dw, float* dbias, long long M, int K, cips_stream_t stream);
/* dx (M,K) = drgb (M,3) @ w (3,K) [+ add]; optional copy before masking; out = dx * (mask>0 ? 1 : slope)
 * (the LeakyReLU gate of the layer below, fused).  mask / add / out_unmasked may be NULL. */
int cips_torgb_bwd_x(const float* drgb, const float* w, const float* add, const float* mask, float slope,
                     float* out_unmasked, float* out, long long M, int K, cips_stream_t stream);
/* the same with split-plane output and the gate as a bit plane (bit k&7 of byte [m][k>>3]; NULL: none):
 * p_hi / p_lo (M, K) bf16 planes of (drgb @ w) * (bit ? 1 : slope); out_unmasked: optional fp32 copy before gating.  K % 8 == 0. */
int cips_torgb_bwd_x_x3(const float* drgb, const float* w, const void* gate_bits, float slope, float* out_unmasked,
                        void* p_hi, void* p_lo, long long M, int K, cips_stream_t stream);

/* Camera pose of a batch (exp/comm/comm_utils.py:451-581: sample_camera_positions for 'gaussian' / 'normal' (uniform = 0:
 * angle = draw * stddev + mean) and 'uniform' (uniform = 1: (draw - 0.5) * 2 * stddev + mean), camera_origin_from_angles with
 * r = 1, look-at-origin forward vector, create_cam2world_matrix with up = (0, 1, 0)) in one launch:
 * theta_raw, phi_raw (B) raw draws -> pitch_yaw (B, 2) = (clamped phi, theta), origin (B, 3), cam2world (B, 4, 4). */
int cips_camera_pose(const float* theta_raw, const float* phi_raw, int uniform, float h_stddev, float h_mean,
                     float v_stddev, float v_mean, float* pitch_yaw, float* origin, float* cam2world, int B,
                     cips_stream_t stream);

/* ------------------------------------------------------------------ */
/* H5  discriminator native ops                                        */
/* ------------------------------------------------------------------ */
/* Same contract as the reference's pybind op
 *   fused.fused_bias_act(input, bias, refer, act, grad, alpha, scale)
 * (exp/comm/op/fused_bias_act.cpp:11-21, kernel fused_bias_act_kernel.cu:18-49):
 * y = f(x + b[(i/step_b) % size_b]) * scale, act*10+grad switch:
 * 10 linear, 11/12 linear grads, 30 lrelu, 31 lrelu grad gated by sign(refer),
 * 32 second-order (=0).  bias / refer may be NULL ("empty tensor"). */
int cips_fused_bias_act(const float* x, const float* bias, const float* refer, float* y,
                        long long numel, int size_b, int step_b,
                        int act, int grad, float alpha, float scale, cips_stream_t stream);

/* Same contract as upfirdn2d_op.upfirdn2d(input[N,H,W,minor], kernel[kh,kw],
 * up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1)
 * (exp/comm/op/upfirdn2d.cpp:12-23, kernel upfirdn2d_kernel.cu:52-137). */
int cips_upfirdn2d(const float* input, const float* kernel, float* out,
                   int major, int in_h, int in_w, int minor, int kernel_h, int kernel_w,
                   int up_x, int up_y, int down_x, int down_y,
                   int pad_x0, int pad_x1, int pad_y0, int pad_y1, cips_stream_t stream);
/* The same op for a 4 x 4 kernel, up = down = 1, on `major` planes of in_h x in_w that are stored as the four parity blocks
 * written by cips_conv2d_x3_dgrad_s2 (pixel (y, x) of plane m at dxp[blk_off[2(y&1) + (x&1)] + m * Np + (y>>1) * Ws + (x>>1)],
 * Ws = ceil((in_w - (x&1)) / 2), Np = Hs * Ws rounded up to 8): the transpose of the Blur of a down-sampling ConvLayer
 * (discriminator.py:57-82, 190-203) applied to the stride-2 convolution's data gradient without materialising it
 * row-major.  Same taps in the same order as cips_upfirdn2d on the interleaved tensor: bit-identical results. */
int cips_upfirdn2d_parity(const float* dxp, const long long* blk_off, const float* kernel, float* out, int major,
                          int in_h, int in_w, int pad_x0, int pad_x1, int pad_y0, int pad_y1, cips_stream_t stream);

/* EqualLinear (exp/cips3d/models/discriminator.py:254-288: F.linear(input, weight * scale) [+ bias * lr_mul]) and the two
 * other bilinear forms of its autograd (each form's gradients are the other two: the R1 double-backward closes):
 *   mode 0   out (B, O) = s * a (B, K) . b^T (O, K)  [+ bias (O) * bias_scale]        a = x, b = weight
 *   mode 1   out (B, K) = s * a (B, O) . b (O, K)                                      a = dy, b = weight
 *   mode 2   out (O, K) = s * a^T (O, B) . b (B, K)                                    a = dy, b = x
 * O % 4 == 0 and K % 4 == 0: on cips_gemm_f32 (mode 0 with K >= 2048 cut into 512-wide chunks whose partial products
 * land in `scratch`, cips_equal_linear_scratch floats, and are summed in chunk order); else three streaming kernels
 * (the 512 -> 1 output layer).  bias only with mode 0. */
long long cips_equal_linear_scratch(int mode, int B, int K, int O);
int cips_equal_linear(int mode, const float* a, const float* b, const float* bias, float bias_scale, float s, float* out,
                      float* scratch, int B, int K, int O, cips_stream_t stream);

/* DiffAugment with policy 'color,translation,cutout' (exp/cips3d/models/diffaug.py:9-85; applied to every discriminator
 * input, exp/cips3d/models/discriminator.py:507-508) as one affine operator and its adjoint.  rb, rs, rc: the (B) raw
 * uniform draws of brightness / saturation / contrast; tx, ty, ox, oy: the (B) int64 draws of translation and cutout
 * centre, all made by the host with the reference's calls.  x, y (B, C <= 4, H, W); sums: 33 * B floats of scratch
 * (per-image sums, then 32 slices of partial sums per image).
 * adjoint 0: y = A x + c (affine != 0) or y = A x (affine == 0); adjoint 1: y = A^T x (the backward; its own backward
 * is the forward with affine == 0: the R1 double-backward of train.py:387-394).  cut_h / cut_w = int(size * 0.2 + 0.5).
 * Policies that leave stages out (diffaug.py:12-17 applies the listed ones): bit 1 of `affine` set = no colour stage
 * (values pass through untouched), tx = ty = 0 = no translation, cut_h = cut_w = 0 = no cutout. */
int cips_diffaug(const float* x, float* y, const float* rb, const float* rs, const float* rc, const long long* tx,
                 const long long* ty, const long long* ox, const long long* oy, float* sums, int B, int C, int H, int W,
                 int cut_h, int cut_w, int adjoint, int affine, cips_stream_t stream);
/* Progressive fade-in (discriminator.py:524-534): the 2x2 mean that F.interpolate(scale_factor=0.5, 'bilinear') computes
 * on even sizes (adjoint 1: its transpose, 0.25 g to the four sources), and out = a x + b y (y may be NULL). */
int cips_avgpool2(const float* x, float* y, long long planes, int H, int W, int adjoint, cips_stream_t stream);
int cips_axpby(const float* x, const float* y, float* out, float a, float b, long long n, cips_stream_t stream);

/* FID path: float image (B, C <= 4, H, W) -> uint8 pixels (B, H, W, C) exactly as torchvision.utils.save_image(img,
 * normalize=True, value_range=(lo, hi)) quantises them before JPEG encoding (exp/cips3d/scripts/gen_images.py:56-60;
 * torchvision/utils.py make_grid norm_ip + save_image): clamp to [lo, hi], (x - lo) * (1 / max(hi - lo, 1e-5)),
 * * 255, + 0.5, clamp to [0, 255], truncate.  Bit-exact on identical float inputs. */
int cips_image_to_u8(const float* x, unsigned char* out, int B, int C, int H, int W, float lo, float hi,
                     cips_stream_t stream);

/* Backward of FusedLeakyReLU on a (B, C, H, W) activation (exp/comm/op/fused_act.py:26-44) with the bias gradient in the
 * same pass: grad_in = (refer > 0 ? grad : grad * alpha) * scale — cips_fused_bias_act(act 3, grad 1) — and
 * part[plane][slice] = sum of grad_in over slice `slice` of plane (b, c); planes = B * C, slices =
 * cips_lrelu_bwd_bias_slices(HW); the caller adds part over images and slices to get grad_bias (C). */
int cips_lrelu_bwd_bias_slices(int HW);
int cips_lrelu_bwd_bias(const float* grad, const float* refer, float* grad_in, float* part, long long planes, int HW,
                        float alpha, float scale, cips_stream_t stream);
/* grad_bias[c] = sum over b < B, s < S of part[b][c][s] (fixed order): the tail of cips_lrelu_bwd_bias, planes = B * C */
int cips_lrelu_bwd_bias_finish(const float* part, float* grad_bias, int B, int C, int S, cips_stream_t stream);
/* cips_lrelu_bwd_bias written directly as the NHWC split planes (B*HW + 1, C; zero last row) of the gated gradient — for
 * convolutions that read their incoming gradient only through those planes the fp32 tensor never exists.  part: (B, C,
 * cips_lrelu_bwd_bias_nhwc_tiles(HW)) partial bias sums for cips_lrelu_bwd_bias_finish.  C % 8 == 0.  Plane values are those of
 * cips_lrelu_bwd_bias followed by cips_split_planes_nhwc, bit for bit. */
int cips_lrelu_bwd_bias_nhwc_tiles(int HW);
int cips_lrelu_bwd_bias_nhwc(const float* grad, const float* refer, void* t_hi, void* t_lo, float* part, int B, int C, int HW,
                             float alpha, float scale, cips_stream_t stream);

/* 1x1 convolution with C <= 4 input channels (EqualConv2d of the RGB input layers, discriminator.py:457-459):
 * y (B, O, HW) = w (O, C) . x (B, C, HW); HW % 4 == 0.  Streaming kernel, no GEMM. */
int cips_conv1x1_smallk(const float* x, const float* w, float* y, int B, int C, int O, int HW, cips_stream_t stream);
/* its data gradient: dx (B, C, HW) = w^T (C, O) . dy (B, O, HW) */
int cips_conv1x1_smallk_bwd_data(const float* dy, const float* w, float* dx, int B, int C, int O, int HW,
                                 cips_stream_t stream);
/* its weight gradient (autograd of discriminator.py:44-48 for the RGB layers): part (S, O, C), S =
 * cips_conv1x1_smallk_bwd_weight_splits(O, HW) partial rows that the caller sums over S:
 * sum_s part[s][o][c] = sum_{b,p} dy[b][o][p] x[b][c][p] */
int cips_conv1x1_smallk_bwd_weight_splits(int O, int HW);
int cips_conv1x1_smallk_bwd_weight(const float* dy, const float* x, float* part, int B, int C, int O, int HW,
                                   cips_stream_t stream);

/* im2col for the EqualConv2d GEMM path (exp/cips3d/models/discriminator.py:40-48).
 * x (B,C,H,W) NCHW -> col (B, C*kh*kw, Ho*Wo) row-major ("colT": k-major B operand, so that
 * out[b] (O, Ho*Wo) = W (O, C*kh*kw) @ col[b] lands directly in NCHW).  col2im is the adjoint
 * (gather form, deterministic). */
int cips_im2col(const float* x, float* col, int B, int C, int H, int W,
                int kh, int kw, int stride, int pad, cips_stream_t stream);
/* im2col writing split-bf16 planes (col = hi + lo), the operand form of the bf16x3 GEMMs */
int cips_im2col_x3(const float* x, void* col_hi, void* col_lo, int B, int C, int H, int W, int kh, int kw,
                   int stride, int pad, cips_stream_t stream);
int cips_col2im(const float* col, float* dx, int B, int C, int H, int W,
                int kh, int kw, int stride, int pad, cips_stream_t stream);

/* ------------------------------------------------------------------ */
/* Training-step tail: gradient-norm clip + Adam + EMA over all tensors of one optimiser,
 * two launches.  Replaces torch.nn.utils.clip_grad_norm_ + torch.optim.Adam.step + EMA.update
 * (exp/cips3d/scripts/train.py:420-491, exp/comm/comm_model_utils.py:97-118).
 * table_dev: device array of tensors; grad == NULL: no Adam update (parameter unused this step), ema == NULL: no EMA.
 * The tensors are walked in chunks of cips_opt_chunk() elements: chunk_tensor_dev[c] = tensor index,
 * chunk_off_dev[c] = first element; partial_dev: nchunks doubles of scratch; total_norm_dev (optional): pre-clip norm.
 * max_norm <= 0 disables clipping; `step` = this tensor's 1-based Adam step count (torch counts per parameter);
 * write_grad != 0 stores the clipped gradient back.
 * steps_dev (optional, one long long per tensor): device-side step counts — the call advances the count of every
 * tensor that has a gradient and uses it instead of the table's `step`, so neither the table nor a captured hipGraph of
 * the call goes stale from one step to the next. */
typedef struct cips_opt_tensor {
  float* param; const float* grad; float* exp_avg; float* exp_avg_sq; float* ema; long long n; long long step;
} cips_opt_tensor;
int cips_opt_chunk(void);
int cips_opt_step(const cips_opt_tensor* table_dev, const int* chunk_tensor_dev, const long long* chunk_off_dev,
                  int nchunks, double* partial_dev, float* total_norm_dev, float max_norm, float lr,
                  float beta1, float beta2, float eps, float ema_decay, int write_grad,
                  long long* steps_dev, cips_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CIPS3D_HIP_H */
