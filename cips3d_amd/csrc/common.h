// common.h — shared device helpers for the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CIPS_CHECK_LAUNCH() (int)hipGetLastError()

// Tuning aids (phase skipping, store suppression, start-phase skew, in-kernel timestamps) produce WRONG results by design and
// exist only in probe builds: `hipcc -DCIPS_TUNING` (scripts/probe/build_tuning.sh).  The production library compiles them
// out — CIPS_TUNE(x) is the constant 0 — and reads no environment variable at all (tests/test_abi.py checks the binary).
#ifdef CIPS_TUNING
#include <stdlib.h>
#define CIPS_TUNE(x) (x)
static inline int cips_tune_env(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#else
#define CIPS_TUNE(x) 0
#endif

// cross-file entry points that are not part of the C-ABI (include/cips3d_hip.h): kept out of the dynamic symbol table
#define CIPS_INTERNAL __attribute__((visibility("hidden")))

// Launchers cache per-DEVICE facts in function-local statics (dynamic-LDS attribute set, CU count).  One process
// normally drives one GPU, but nothing enforces it: a cached flag is reset whenever the calling thread's current device
// is not the one it was set for, so a second GPU gets its own hipFuncSetAttribute / CU count.
static inline int cips_current_device() { int d = 0; (void)hipGetDevice(&d); return d; }
#define CIPS_PER_DEVICE(flag, zero)                                                      \
  do { static int cips_dev_ = -1; const int cips_d_ = cips_current_device();             \
       if (cips_d_ != cips_dev_) { flag = zero; cips_dev_ = cips_d_; } } while (0)

// wave-wide sum without LDS or index registers: DPP row / bank steps in a fixed order, the total read from lane 63
#define CIPS_DPP_ADD(v, ctrl, rmask) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rmask, 0xF, false))
__device__ __forceinline__ float wave_sum_dpp(float v) {
  CIPS_DPP_ADD(v, 0xB1, 0xF);       // quad_perm [1,0,3,2]
  CIPS_DPP_ADD(v, 0x4E, 0xF);       // quad_perm [2,3,0,1]
  CIPS_DPP_ADD(v, 0x141, 0xF);      // row_half_mirror
  CIPS_DPP_ADD(v, 0x140, 0xF);      // row_mirror: every lane of a row holds the row's sum
  CIPS_DPP_ADD(v, 0x142, 0xA);      // row_bcast:15 into rows 1 and 3
  CIPS_DPP_ADD(v, 0x143, 0xC);      // row_bcast:31 into rows 2 and 3: lane 63 holds the wave's sum
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Row (within a 32x32 MFMA C/D tile) held by accumulator register r of a lane
// whose upper-half flag is hf = lane >> 5.  Column = lane & 31.
// (cdna_hip_programming.md §3: row=(reg&3)+8*(reg>>2)+4*(lane>>5))
__device__ __forceinline__ constexpr int mfma_row(int r, int hf) {
  return (r & 3) + 8 * (r >> 2) + 4 * hf;
}

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float lrelu(float x, float slope) { return x > 0.f ? x : x * slope; }

// sin / cos with explicit 2-term Cody-Waite range reduction, then the hardware
// v_sin_f32 / v_cos_f32 (argument in revolutions).  FiLM gains are ~30, so
// arguments reach tens of radians; reduce in fp32 first.
#define CIPS_INV_2PI 0.15915494309189535f
#define CIPS_2PI_HI 6.2831854820251465f
#define CIPS_2PI_LO (-1.7484555314695172e-7f)

__device__ __forceinline__ float reduce_2pi(float x) {
  float k = rintf(x * CIPS_INV_2PI);
  float r = fmaf(-k, CIPS_2PI_HI, x);
  r = fmaf(-k, CIPS_2PI_LO, r);
  return r;  // in [-pi, pi] (+- rounding)
}

// Polynomial sin/cos on the reduced argument r in [-pi, pi]: fold to
// [-pi/4, pi/4] by quadrant and evaluate the classic Cephes sinf/cosf minimax
// polynomials (abs error ~1e-7).
__device__ __forceinline__ void sincos_reduced(float r, float* s, float* c) {
  float q = rintf(r * 0.6366197723675814f);  // 2/pi
  float t = fmaf(-q, 1.5707963705062866f, r);
  t = fmaf(-q, -4.371138828673793e-8f, t);
  int qi = (int)q;
  float z = t * t;
  float sp = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
  sp = fmaf(sp, z, -1.6666654611e-1f);
  float st = fmaf(sp * z, t, t);
  float cp = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
  cp = fmaf(cp, z, 4.166664568298827e-2f);
  float ct = fmaf(cp, z * z, fmaf(-0.5f, z, 1.0f));
  float ss, cc;
  if (qi & 1) { ss = ct; cc = -st; } else { ss = st; cc = ct; }
  if (qi & 2) { ss = -ss; cc = -cc; }
  *s = ss; *c = cc;
}

template <bool HW>
__device__ __forceinline__ float film_sin(float x) {
  float r = reduce_2pi(x);
  if (HW) return __builtin_amdgcn_sinf(r * CIPS_INV_2PI);
  float s, c; sincos_reduced(r, &s, &c); return s;
}
// Forward chain of the split-bf16 kernels: one multiply to revolutions + v_fract_f32 + v_sin_f32 (3 instructions
// instead of 6).  |arg| is tens of radians: the product's rounding is <= 6e-6 rad, an order of magnitude below what the
// 3-pass bf16 pre-activations carry (~5e-6 relative of arguments that large); the exact-fp32 kernels keep the
// two-term Cody-Waite reduction.
template <bool HW>
__device__ __forceinline__ float film_sin_x3(float x) {
  if (HW) return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(x * CIPS_INV_2PI));
  float s, c; sincos_reduced(reduce_2pi(x), &s, &c); return s;
}
template <bool HW>
__device__ __forceinline__ void film_sincos(float x, float* s, float* c) {
  float r = reduce_2pi(x);
  if (HW) {
    float rv = r * CIPS_INV_2PI;
    *s = __builtin_amdgcn_sinf(rv);
    *c = __builtin_amdgcn_cosf(rv);
  } else {
    sincos_reduced(r, s, c);
  }
}
