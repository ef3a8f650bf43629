// optim.hip — multi-tensor training-step tail for gfx950: global gradient-norm clip + Adam + EMA in two launches.
//
// Replaces, for one optimiser (exp/cips3d/scripts/train.py:420-491, exp/comm/comm_model_utils.py:97-118):
//   total_norm = clip_grad_norm_(params, max_norm)            (norm of the per-tensor 2-norms, coef = max/(norm+1e-6) <= 1)
//   Adam.step()  (torch.optim.Adam, amsgrad off, weight_decay 0: lerp of exp_avg, mul/addcmul of exp_avg_sq,
//                 bias corrections from the step count, addcdiv into the parameter)
//   EMA.update() target = target * decay + source * (1 - decay)
// which the reference runs as ~10 tiny elementwise kernels per tensor over ~170 (G) / ~160 (D) tensors plus a host
// sync in .item().  Here the tensors are described by one device table and walked in 64K-element chunks:
//   launch 1: per-chunk sum of squares of the gradients (double accumulation across threads, fixed order)
//   launch 2: every workgroup re-reduces the chunk partials in the same fixed order (deterministic, no atomics, no
//             host round trip), forms the clip coefficient and updates its chunk.
// HBM-bound streaming: 7 floats read + 4 written per parameter.
#include "common.h"
#include "../../include/cips3d_hip.h"

namespace {

constexpr int CHUNK = 65536;

__global__ __launch_bounds__(256) void opt_sqnorm_kernel(const cips_opt_tensor* __restrict__ T,
                                                         const int* __restrict__ chunk_tensor,
                                                         const long long* __restrict__ chunk_off,
                                                         double* __restrict__ partial, long long* __restrict__ steps) {
  __shared__ double red[256];
  const int c = blockIdx.x;
  const cips_opt_tensor t = T[chunk_tensor[c]];
  const long long off = chunk_off[c];
  // device-side Adam step counts (per tensor, like torch.optim.Adam's state): the first chunk of a tensor that has a
  // gradient advances it here; the update launch reads it.  Nothing about the step count is baked into host-written
  // data, so a captured hipGraph of the step advances its bias corrections on every replay.
  if (steps && t.grad && off == 0 && threadIdx.x == 0) steps[chunk_tensor[c]] += 1;
  const long long end = (off + CHUNK < t.n) ? off + CHUNK : t.n;
  float acc = 0.f;
  if (t.grad)
    for (long long i = off + threadIdx.x; i < end; i += 256) { const float g = t.grad[i]; acc = fmaf(g, g, acc); }
  red[threadIdx.x] = (double)acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[c] = red[0];
}

struct StepArgs {
  const cips_opt_tensor* T;
  const int* chunk_tensor;
  const long long* chunk_off;
  const double* partial;
  const long long* steps;     // optional device-side per-tensor step counts (already advanced by launch 1)
  float* total_norm;          // optional: the pre-clip norm, for logging
  int nchunks;
  float max_norm;             // <= 0: no clipping
  float lr, beta1, beta2, eps;
  float ema_decay;
  int write_grad;             // clip_grad_norm_ scales .grad in place: keep that visible
};

__global__ __launch_bounds__(256) void opt_step_kernel(StepArgs a) {
  __shared__ double red[256];
  __shared__ float s_coef;
  // total norm: every workgroup sums all chunk partials in the same order
  double acc = 0.0;
  for (int i = threadIdx.x; i < a.nchunks; i += 256) acc += a.partial[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(red[0]);
    float coef = 1.f;
    if (a.max_norm > 0.f) { coef = a.max_norm / (norm + 1e-6f); coef = coef > 1.f ? 1.f : coef; }
    s_coef = coef;
    if (blockIdx.x == 0 && a.total_norm) *a.total_norm = norm;
  }
  __syncthreads();
  const float coef = s_coef;
  const int c = blockIdx.x;
  const cips_opt_tensor t = a.T[a.chunk_tensor[c]];
  const long long off = a.chunk_off[c];
  const long long end = (off + CHUNK < t.n) ? off + CHUNK : t.n;
  // bias corrections from THIS tensor's step count (torch.optim.Adam counts steps per parameter: one that had no
  // gradient in some iteration lags behind)
  long long step = a.steps ? a.steps[a.chunk_tensor[c]] : t.step;
  step = step < 1 ? 1 : step;
  const float bc1 = (float)(1.0 - pow((double)a.beta1, (double)step));
  const float bc2_sqrt = (float)sqrt(1.0 - pow((double)a.beta2, (double)step));
  const float step_size = a.lr / bc1;
  for (long long i = off + threadIdx.x; i < end; i += 256) {
    float p = t.param[i];
    if (t.grad) {
      const float g = t.grad[i] * coef;
      if (a.write_grad) const_cast<float*>(t.grad)[i] = g;
      float m = t.exp_avg[i], v = t.exp_avg_sq[i];
      const float w = 1.f - a.beta1;                                        // exp_avg.lerp_(grad, 1 - beta1)
      m = (w < 0.5f) ? m + w * (g - m) : g - (g - m) * (1.f - w);
      v = v * a.beta2 + ((1.f - a.beta2) * g) * g;                          // mul_(beta2).addcmul_(g, g, 1 - beta2)
      const float denom = sqrtf(v) / bc2_sqrt + a.eps;
      p = p + (-step_size) * (m / denom);                                   // addcdiv_
      t.exp_avg[i] = m; t.exp_avg_sq[i] = v;
      t.param[i] = p;
    }
    if (t.ema) t.ema[i] = t.ema[i] * a.ema_decay + p * (1.f - a.ema_decay);
  }
}

}  // namespace

extern "C" int cips_opt_chunk(void) { return CHUNK; }

extern "C" int cips_opt_step(const cips_opt_tensor* table_dev, const int* chunk_tensor_dev, const long long* chunk_off_dev,
                             int nchunks, double* partial_dev, float* total_norm_dev, float max_norm, float lr,
                             float beta1, float beta2, float eps, float ema_decay, int write_grad,
                             long long* steps_dev, cips_stream_t stream) {
  if (!table_dev || !chunk_tensor_dev || !chunk_off_dev || !partial_dev || nchunks <= 0) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(opt_sqnorm_kernel, dim3(nchunks), dim3(256), 0, st, table_dev, chunk_tensor_dev, chunk_off_dev, partial_dev, steps_dev);
  StepArgs a;
  a.T = table_dev; a.chunk_tensor = chunk_tensor_dev; a.chunk_off = chunk_off_dev; a.partial = partial_dev; a.steps = steps_dev;
  a.total_norm = total_norm_dev; a.nchunks = nchunks; a.max_norm = max_norm;
  a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
  a.ema_decay = ema_decay; a.write_grad = write_grad;
  hipLaunchKernelGGL(opt_step_kernel, dim3(nchunks), dim3(256), 0, st, a);
  return CIPS_CHECK_LAUNCH();
}
