// disc_ops.hip — the discriminator's native ops for gfx950 (H5): the two ops the reference
// ships as CUDA extensions, with identical contracts, plus the im2col / col2im pair that
// feeds EqualConv2d to the fp32 MFMA GEMM.  All HBM-bandwidth bound streaming kernels.
#include "common.h"
#include "../../include/cips3d_hip.h"

// A readable zero outside any tensor: out-of-range taps of the register-tiled upfirdn2d kernels are redirected here by
// ADDRESS (value selects let hipcc sink each load into its own predicated block with a full wait behind it: 49 serial
// round trips per thread).  External linkage on purpose, so that the loads cannot be folded to a constant.
__device__ float cips_zero_word[4];

namespace {

// exp/comm/op/fused_bias_act_kernel.cu:18-49
__global__ __launch_bounds__(256) void fused_bias_act_kernel(const float* __restrict__ x, const float* __restrict__ b,
                                                             const float* __restrict__ ref, float* __restrict__ y,
                                                             long long n, int size_b, int step_b, int mode,
                                                             float alpha, float scale) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float v = x[i];
    if (b) v += b[(i / step_b) % size_b];
    const float r = ref ? ref[i] : 0.f;
    float o;
    switch (mode) {
      default:
      case 10: o = v; break;
      case 11: o = v; break;
      case 12: o = 0.f; break;
      case 30: o = (v > 0.f) ? v : v * alpha; break;
      case 31: o = (r > 0.f) ? v : v * alpha; break;
      case 32: o = 0.f; break;
    }
    y[i] = o * scale;
  }
}

// The same op on whole planes: blockIdx.y walks the planes (a plane = step_b contiguous elements that share one bias
// value, i.e. one (image, channel) map), threads take float4s of the plane — no per-element 64-bit division / modulo, 16-byte
// accesses.  Bit-identical results (the arithmetic per element is unchanged).  Needs step_b % 4 == 0 and 16-byte aligned
// tensors; every feature map of the discriminator qualifies, the (rows, C) outputs of the linear layers (step_b = 1) and odd
// planes take the scalar kernel above.
__device__ __forceinline__ float fba_one(float v, float r, int mode, float alpha) {
  switch (mode) {
    default:
    case 10: return v;
    case 11: return v;
    case 12: return 0.f;
    case 30: return (v > 0.f) ? v : v * alpha;
    case 31: return (r > 0.f) ? v : v * alpha;
    case 32: return 0.f;
  }
}
__global__ __launch_bounds__(256) void fused_bias_act_planes_kernel(const float4* __restrict__ x, const float* __restrict__ b,
                                                                    const float4* __restrict__ ref, float4* __restrict__ y,
                                                                    long long planes, int size_b, int step4, int mode,
                                                                    float alpha, float scale) {
  for (long long pl = blockIdx.y; pl < planes; pl += gridDim.y) {
    const float bias = b ? b[pl % size_b] : 0.f;
    const long long base = pl * step4;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < step4; i += gridDim.x * 256) {
      float4 v = x[base + i];
      v.x += bias; v.y += bias; v.z += bias; v.w += bias;
      float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ref) r = ref[base + i];
      float4 o;
      o.x = fba_one(v.x, r.x, mode, alpha) * scale; o.y = fba_one(v.y, r.y, mode, alpha) * scale;
      o.z = fba_one(v.z, r.z, mode, alpha) * scale; o.w = fba_one(v.w, r.w, mode, alpha) * scale;
      y[base + i] = o;
    }
  }
}

__device__ __forceinline__ int floor_div(int a, int b) {
  int c = a / b;
  if (c * b > a) c--;
  return c;
}

// exp/comm/op/upfirdn2d_kernel.cu:52-137 — same arithmetic (polyphase, flipped kernel), one
// thread per output element; the <= kh*kw input taps come from L1/L2 (each input element is
// reused by kh*kw/(down^2) outputs of neighbouring lanes), so HBM sees input once + output once.
struct UpfirArgs {
  const float *in, *k;
  float* out;
  int major, in_h, in_w, minor, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_y0, out_h, out_w;
};

__global__ __launch_bounds__(256) void upfirdn2d_kernel(UpfirArgs a) {
  __shared__ float sk[64];
  for (int t = threadIdx.x; t < a.kh * a.kw; t += 256) {
    int ky = t / a.kw, kx = t % a.kw;
    sk[t] = a.k[(a.kh - 1 - ky) * a.kw + (a.kw - 1 - kx)];
  }
  __syncthreads();
  const long long total = (long long)a.major * a.out_h * a.out_w * a.minor;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int mi = (int)(idx % a.minor);
    long long t = idx / a.minor;
    const int ox = (int)(t % a.out_w); t /= a.out_w;
    const int oy = (int)(t % a.out_h);
    const long long mj = t / a.out_h;
    const int mid_x = ox * a.down_x + a.up_x - 1 - a.pad_x0;
    const int mid_y = oy * a.down_y + a.up_y - 1 - a.pad_y0;
    const int in_x0 = floor_div(mid_x, a.up_x), in_y0 = floor_div(mid_y, a.up_y);
    const int kx0 = (in_x0 + 1) * a.up_x - mid_x - 1, ky0 = (in_y0 + 1) * a.up_y - mid_y - 1;
    float v = 0.f;
    for (int yy = 0, ky = ky0; ky < a.kh; ++yy, ky += a.up_y) {
      const int iy = in_y0 + yy;
      if (iy < 0 || iy >= a.in_h) continue;
      for (int xx = 0, kx = kx0; kx < a.kw; ++xx, kx += a.up_x) {
        const int ix = in_x0 + xx;
        if (ix < 0 || ix >= a.in_w) continue;
        v += a.in[((mj * a.in_h + iy) * a.in_w + ix) * a.minor + mi] * sk[ky * a.kw + kx];
      }
    }
    a.out[idx] = v;
  }
}

// Fast path of the same op for what the discriminator's Blur actually asks for (discriminator.py:57-82): no
// upsampling, minor == 1, a kernel of at most 4 x 4 taps, non-negative padding, down 1 or 2.  The generic kernel above
// runs at 2.3 TB/s on the 64x64x512-channel maps (four 64-bit divisions and up to 16 bounds-checked loads per
// output); here a workgroup stages a band of input rows of one plane in LDS with the zero padding materialised
// (16-byte global loads), every thread produces strips of four neighbouring outputs from registers (7 or 10 LDS values
// per kernel row instead of 16 loads per output), and stores them coalesced.  Same taps in the same order (rows
// outer, columns inner; padding contributes exact zeros), so the results are those of the generic kernel.
// Register-tiled form for the 4 x 4 blur the discriminator uses everywhere (up 1, down 1 or 2): a thread produces a 4 x 4
// block of outputs straight from global memory — (3 DOWN + 4)^2 clamped dword loads that neighbouring lanes share through
// L1 (7 x 7 for 16 outputs at down 1), 256 FMAs, no LDS, no barriers, no per-element divisions (two per 16 outputs).  The
// LDS-band kernel below spends ~30 instructions of index arithmetic per staged element and serialises stage / barrier /
// compute per plane: 125 us on the 8192 64 x 64 planes whose bytes need 45, 117 us on down-2 calls that need 30.  Same
// taps in the same order (kernel rows outer, columns inner, one fma chain per output), so the results are bit-identical
// to the generic kernel's.
// Thread tile TW x TH outputs.  4 x 4 minimises loads per output (3.1) but neighbouring lanes then read addresses 16 B apart:
// every wave load touches 1 KiB for 256 useful bytes and the texture path moves 12.5 x the output bytes.  1 x 16 (lanes
// along x) needs 4.75 loads per output, each fully coalesced: 4.75 x.
// PAR: the input plane is not stored row-major but as the four parity blocks cips_conv2d_x3_dgrad_s2 writes — pixel (iy, ix) of
// plane mj lives at blk[iy&1][ix&1] + mj * np[..] + (iy>>1) * ws[ix&1] + (ix>>1).  Only the load address changes: same taps, same order.
struct ParityIn { const float* blk[4]; int np[4]; int ws[2]; };
template <int DOWN, int TW, int TH, bool PAR = false>
__global__ __launch_bounds__(256) void upfirdn2d_direct_kernel(UpfirArgs a, int bw, int bh, ParityIn pin = ParityIn()) {
  constexpr int NX = (TW - 1) * DOWN + 4, NR = (TH - 1) * DOWN + 4;
  float ck[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) ck[i] = a.k[15 - i];                   // flipped 4 x 4 kernel (uniform: scalar loads)
  const long long per_plane = (long long)bh * bw, nblk = a.major * per_plane;
  const bool vec = TW == 4 && (a.out_w & 3) == 0;
  const float* zp = cips_zero_word;
  for (long long t = blockIdx.x * 256LL + threadIdx.x; t < nblk; t += gridDim.x * 256LL) {
    const long long mj = t / per_plane;
    const int rem = (int)(t - mj * per_plane), by = rem / bw, bx = rem - by * bw;
    const int oy0 = by * TH, ox0 = bx * TW;
    const int iyb = oy0 * DOWN - a.pad_y0, ixb = ox0 * DOWN - a.pad_x0;
    const float* src = PAR ? nullptr : a.in + mj * (long long)a.in_h * a.in_w;
    int colx[NX];
    bool cok[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int ix = ixb + i;
      cok[i] = (unsigned)ix < (unsigned)a.in_w;
      colx[i] = min(max(ix, 0), a.in_w - 1);
    }
    float v[TH][TW];
#pragma unroll
    for (int ry = 0; ry < TH; ++ry)
#pragma unroll
      for (int o = 0; o < TW; ++o) v[ry][o] = 0.f;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int iy = iyb + r;
      const bool rok = (unsigned)iy < (unsigned)a.in_h;
      float x[NX];
      if constexpr (PAR) {
        // (no dynamic index into the by-value argument struct: that would move it to scratch)
        const int iyc = min(max(iy, 0), a.in_h - 1), U = iyc >> 1;
        const bool odd = iyc & 1;
        const float* p0 = (odd ? pin.blk[2] : pin.blk[0]) + mj * (long long)(odd ? pin.np[2] : pin.np[0]) + (long long)U * pin.ws[0];
        const float* p1 = (odd ? pin.blk[3] : pin.blk[1]) + mj * (long long)(odd ? pin.np[3] : pin.np[1]) + (long long)U * pin.ws[1];
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = *((rok && cok[i]) ? ((colx[i] & 1) ? p1 : p0) + (colx[i] >> 1) : zp);
      } else {
        const float* rowp = src + (long long)min(max(iy, 0), a.in_h - 1) * a.in_w;
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = *((rok && cok[i]) ? rowp + colx[i] : zp);
      }
#pragma unroll
      for (int ry = 0; ry < TH; ++ry) {
        const int ky = r - ry * DOWN;
        if (ky >= 0 && ky < 4) {
#pragma unroll
          for (int kx = 0; kx < 4; ++kx)
#pragma unroll
            for (int o = 0; o < TW; ++o) v[ry][o] = fmaf(x[o * DOWN + kx], ck[ky * 4 + kx], v[ry][o]);
        }
      }
    }
#pragma unroll
    for (int ry = 0; ry < TH; ++ry) {
      const int oy = oy0 + ry;
      if (oy < a.out_h) {
        float* q = a.out + (mj * a.out_h + oy) * (long long)a.out_w + ox0;
        if constexpr (TW == 4) {
          if (vec) {
            *reinterpret_cast<float4*>(q) = make_float4(v[ry][0], v[ry][1], v[ry][2], v[ry][3]);
            continue;
          }
        }
#pragma unroll
        for (int o = 0; o < TW; ++o)
          if (ox0 + o < a.out_w) q[o] = v[ry][o];
      }
    }
  }
}

// The parity-block form with the lanes along the BLOCK column: thread = output columns 2V and 2V+1 of TH rows.  Its five input
// columns 2V - pad .. 2V - pad + 4 alternate between the two column parities in an order that is the same for every thread, so
// each of the five loads per input row reads ONE block at consecutive addresses (lane = V): fully coalesced, 2.5 loads per
// output and row instead of 4.  (The column-per-lane form above reads both blocks with every instruction: 256 us against
// 127 us for the row-major tensor on the 64 x 64 x 512 gradient.)  Same taps in the same order: bit-identical.
template <int TH>
__global__ __launch_bounds__(256) void upfirdn2d_parity2_kernel(UpfirArgs a, int hw2, int bh, ParityIn pin) {
  float ck[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) ck[i] = a.k[15 - i];
  const long long per_plane = (long long)bh * hw2, nblk = a.major * per_plane;
  const float* zp = cips_zero_word;
  const int par0 = a.pad_x0 & 1;                       // parity of input column ixb + i is (par0 + i) & 1 for every thread
  for (long long t = blockIdx.x * 256LL + threadIdx.x; t < nblk; t += gridDim.x * 256LL) {
    const long long mj = t / per_plane;
    const int rem = (int)(t - mj * per_plane), by = rem / hw2, V = rem - by * hw2;
    const int oy0 = by * TH, ox0 = 2 * V;
    const int iyb = oy0 - a.pad_y0, ixb = ox0 - a.pad_x0;
    int cV[5];
    bool cok[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int ix = ixb + i;
      cok[i] = (unsigned)ix < (unsigned)a.in_w;
      cV[i] = min(max(ix, 0), a.in_w - 1) >> 1;
    }
    float v[TH][2];
#pragma unroll
    for (int ry = 0; ry < TH; ++ry) { v[ry][0] = 0.f; v[ry][1] = 0.f; }
#pragma unroll
    for (int r = 0; r < TH + 3; ++r) {
      const int iy = iyb + r;
      const bool rok = (unsigned)iy < (unsigned)a.in_h;
      const int iyc = min(max(iy, 0), a.in_h - 1), U = iyc >> 1;
      const bool odd = iyc & 1;
      const float* p0 = (odd ? pin.blk[2] : pin.blk[0]) + mj * (long long)(odd ? pin.np[2] : pin.np[0]) + (long long)U * pin.ws[0];
      const float* p1 = (odd ? pin.blk[3] : pin.blk[1]) + mj * (long long)(odd ? pin.np[3] : pin.np[1]) + (long long)U * pin.ws[1];
      const float* pe = par0 ? p1 : p0;                // block of columns i = 0, 2, 4
      const float* po = par0 ? p0 : p1;                // block of columns i = 1, 3
      float x[5];
#pragma unroll
      for (int i = 0; i < 5; ++i) x[i] = *((rok && cok[i]) ? ((i & 1) ? po : pe) + cV[i] : zp);
#pragma unroll
      for (int ry = 0; ry < TH; ++ry) {
        const int ky = r - ry;
        if (ky >= 0 && ky < 4) {
#pragma unroll
          for (int kx = 0; kx < 4; ++kx) {
            v[ry][0] = fmaf(x[kx], ck[ky * 4 + kx], v[ry][0]);
            v[ry][1] = fmaf(x[kx + 1], ck[ky * 4 + kx], v[ry][1]);
          }
        }
      }
    }
#pragma unroll
    for (int ry = 0; ry < TH; ++ry) {
      const int oy = oy0 + ry;
      if (oy < a.out_h) {
        float* q = a.out + (mj * a.out_h + oy) * (long long)a.out_w + ox0;
        if (ox0 + 1 < a.out_w && (a.out_w & 1) == 0) *reinterpret_cast<float2*>(q) = make_float2(v[ry][0], v[ry][1]);
        else { q[0] = v[ry][0]; if (ox0 + 1 < a.out_w) q[1] = v[ry][1]; }
      }
    }
  }
}

// up 2, down 1, 4 x 4 kernel (the backward of the down-2 blur of the skip branch): each output has 2 x 2 taps; the
// quarter-size input stays in L1 / L2.  Same polyphase arithmetic and tap order as the generic kernel.
template <int TH>
__global__ __launch_bounds__(256) void upfirdn2d_up2_kernel(UpfirArgs a, int bh) {
  // thread = one output column x TH output rows, lanes along x (pairs of lanes share their input addresses: coalesced)
  __shared__ float sk[16];
  if (threadIdx.x < 16) sk[threadIdx.x] = a.k[15 - threadIdx.x];
  __syncthreads();
  const long long per_plane = (long long)bh * a.out_w, nblk = a.major * per_plane;
  const float* zp = cips_zero_word;
  for (long long t = blockIdx.x * 256LL + threadIdx.x; t < nblk; t += gridDim.x * 256LL) {
    const long long mj = t / per_plane;
    const int rem = (int)(t - mj * per_plane), by = rem / a.out_w, ox = rem - by * a.out_w;
    const float* src = a.in + mj * (long long)a.in_h * a.in_w;
    const int mid_x = ox + 1 - a.pad_x0, in_x0 = floor_div(mid_x, 2), kx0 = (in_x0 + 1) * 2 - mid_x - 1;
    const bool xok0 = (unsigned)in_x0 < (unsigned)a.in_w, xok1 = (unsigned)(in_x0 + 1) < (unsigned)a.in_w;
    const int xc0 = min(max(in_x0, 0), a.in_w - 1), xc1 = min(max(in_x0 + 1, 0), a.in_w - 1);
    float* dst = a.out + mj * (long long)a.out_h * a.out_w + ox;
#pragma unroll
    for (int j = 0; j < TH; ++j) {
      const int oy = by * TH + j;
      if (oy < a.out_h) {
        const int mid_y = oy + 1 - a.pad_y0, in_y0 = floor_div(mid_y, 2), ky0 = (in_y0 + 1) * 2 - mid_y - 1;
        float acc = 0.f;
#pragma unroll
        for (int yy = 0; yy < 2; ++yy) {
          const int iy = in_y0 + yy, ky = ky0 + 2 * yy;
          const bool rok = (unsigned)iy < (unsigned)a.in_h;
          const float* rowp = src + (long long)min(max(iy, 0), a.in_h - 1) * a.in_w;
          const float q0 = *((rok && xok0) ? rowp + xc0 : zp), q1 = *((rok && xok1) ? rowp + xc1 : zp);
          acc = fmaf(q0, sk[ky * 4 + kx0], acc);
          acc = fmaf(q1, sk[ky * 4 + kx0 + 2], acc);
        }
        dst[(long long)oy * a.out_w] = acc;
      }
    }
  }
}

// Round 6: the two blurs of the skip branch (down 2 forward, up 2 backward) with the lanes along the HALF-resolution column and
// every input element loaded once.  The column-per-lane forms above issue 9 (down 2) and 4 (up 2) dword loads per output whose
// addresses stride by two columns or repeat: 24 / 14 clocks per memory instruction on the texture path, 110 us / 144 us for the
// 64 x 64 x 512 maps whose bytes need ~35.  Here a thread owns output column V (down 2: one 8-byte load of input columns 2V, 2V+1
// per row; up 2: one dword load of input column V per row) and takes columns 2V-1 / 2V+2 (V-1 / V+1) from its neighbour lanes
// by shuffles; a row segment is min(width, 64) lanes, its edges are the image edge (zero padding) or, for rows wider than a
// wave, a single extra load by the edge lane.  Same taps in the same order, one fma chain per output: the values of the
// generic kernel bit for bit.  Shapes: 4 x 4 kernel, pad_x0 = 1 and in_w = 2 out_w (down 2) / pad_x0 = 2 and out_w = 2 in_w
// (up 2), a power-of-two half width; everything else keeps the forms above.
template <int TH>
__global__ __launch_bounds__(256) void upfirdn2d_down2_pairs_kernel(UpfirArgs a, int bh, int seg) {
  float ck[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) ck[i] = a.k[15 - i];
  const long long per_plane = (long long)bh * a.out_w, nblk = a.major * per_plane;
  const int lane = threadIdx.x & 63, ls = lane & (seg - 1);
  for (long long t = blockIdx.x * 256LL + threadIdx.x; t < nblk; t += gridDim.x * 256LL) {
    const long long mj = t / per_plane;
    const int rem = (int)(t - mj * per_plane), by = rem / a.out_w, ox = rem - by * a.out_w;
    const int oy0 = by * TH, iyb = oy0 * 2 - a.pad_y0;
    const float* src = a.in + mj * (long long)a.in_h * a.in_w + 2 * ox;
    float acc[TH];
#pragma unroll
    for (int ry = 0; ry < TH; ++ry) acc[ry] = 0.f;
    constexpr int NR = 2 * (TH - 1) + 4;
    float2 v[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r)                       // every row's load is issued before the first shuffle waits for one
      v[r] = *reinterpret_cast<const float2*>(src + (long long)min(max(iyb + r, 0), a.in_h - 1) * a.in_w);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int iy = iyb + r;
      const bool rok = (unsigned)iy < (unsigned)a.in_h;
      const float* rowp = src + (long long)min(max(iy, 0), a.in_h - 1) * a.in_w;
      if (!rok) v[r] = make_float2(0.f, 0.f);
      float xl = __shfl_up(v[r].y, 1), xr = __shfl_down(v[r].x, 1);
      if (ls == 0) xl = (ox == 0 || !rok) ? 0.f : rowp[-1];
      if (ls == seg - 1) xr = (ox == a.out_w - 1 || !rok) ? 0.f : rowp[2];
      const float x[4] = {xl, v[r].x, v[r].y, xr};
#pragma unroll
      for (int ry = 0; ry < TH; ++ry) {
        const int ky = r - 2 * ry;
        if (ky >= 0 && ky < 4) {
#pragma unroll
          for (int kx = 0; kx < 4; ++kx) acc[ry] = fmaf(x[kx], ck[ky * 4 + kx], acc[ry]);
        }
      }
    }
#pragma unroll
    for (int ry = 0; ry < TH; ++ry) {
      const int oy = oy0 + ry;
      if (oy < a.out_h) a.out[(mj * a.out_h + oy) * (long long)a.out_w + ox] = acc[ry];
    }
  }
}

template <int TU>        // TU input rows = 2 TU output rows per thread
__global__ __launch_bounds__(256) void upfirdn2d_up2_pairs_kernel(UpfirArgs a, int bh, int seg) {
  float ck[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) ck[i] = a.k[15 - i];
  const long long per_plane = (long long)bh * a.in_w, nblk = a.major * per_plane;
  const int lane = threadIdx.x & 63, ls = lane & (seg - 1);
  for (long long t = blockIdx.x * 256LL + threadIdx.x; t < nblk; t += gridDim.x * 256LL) {
    const long long mj = t / per_plane;
    const int rem = (int)(t - mj * per_plane), by = rem / a.in_w, V = rem - by * a.in_w;
    const int U0 = by * TU;
    const float* src = a.in + mj * (long long)a.in_h * a.in_w + V;
    float xl[TU + 2], xc[TU + 2], xr[TU + 2];
#pragma unroll
    for (int i = 0; i < TU + 2; ++i) xc[i] = src[(long long)min(max(U0 - 1 + i, 0), a.in_h - 1) * a.in_w];
#pragma unroll
    for (int i = 0; i < TU + 2; ++i) {
      const int iy = U0 - 1 + i;
      const bool rok = (unsigned)iy < (unsigned)a.in_h;
      const float* rowp = src + (long long)min(max(iy, 0), a.in_h - 1) * a.in_w;
      float c = xc[i];
      if (!rok) c = 0.f;
      float l = __shfl_up(c, 1), r = __shfl_down(c, 1);
      if (ls == 0) l = (V == 0 || !rok) ? 0.f : rowp[-1];
      if (ls == seg - 1) r = (V == a.in_w - 1 || !rok) ? 0.f : rowp[1];
      xl[i] = l; xc[i] = c; xr[i] = r;
    }
    // output row 2U + s_: input rows U - 1 + s_ (kernel row s_) and U + s_ (kernel row s_ + 2); output column 2V: input columns
    // V - 1, V with kernel columns 0, 2; column 2V + 1: input columns V, V + 1 with kernel columns 1, 3  (pad 2 on the low side)
#pragma unroll
    for (int u = 0; u < TU; ++u) {
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) {
        const int oy = 2 * (U0 + u) + s_;
        float e = 0.f, o = 0.f;
#pragma unroll
        for (int yy = 0; yy < 2; ++yy) {
          const int i = u + s_ + yy, ky = s_ + 2 * yy;
          e = fmaf(xl[i], ck[ky * 4 + 0], e); e = fmaf(xc[i], ck[ky * 4 + 2], e);
          o = fmaf(xc[i], ck[ky * 4 + 1], o); o = fmaf(xr[i], ck[ky * 4 + 3], o);
        }
        if (oy < a.out_h) *reinterpret_cast<float2*>(a.out + (mj * a.out_h + oy) * (long long)a.out_w + 2 * V) = make_float2(e, o);
      }
    }
  }
}

// Occupancy: the band lives in DYNAMIC LDS sized to what the launch needs and small planes run in one-wave workgroups —
// with a fixed 32 KiB tile and 256 threads per plane the 32x32 / 16x16 / 8x8 stages (16 384 planes of a few KiB each)
// ran 4 workgroups per CU, one plane each, and took 116 us per call where their bytes need 3-20 us.
constexpr int UF_MAX_LDS_FLOATS = 8192;          // 32 KiB band
template <int DOWN>
__global__ __launch_bounds__(256) void upfirdn2d_blur_kernel(UpfirArgs a, int band_rows, int bands, int lds_w) {
  extern __shared__ __attribute__((aligned(16))) float uf_smem[];
  float* sk = uf_smem;
  float* tile = uf_smem + 16;
  const int nthr = blockDim.x;
  if (threadIdx.x < 16) {
    const int ky = threadIdx.x >> 2, kx = threadIdx.x & 3;
    sk[threadIdx.x] = (ky < a.kh && kx < a.kw) ? a.k[(a.kh - 1 - ky) * a.kw + (a.kw - 1 - kx)] : 0.f;
  }
  const int in_rows_band = (band_rows - 1) * DOWN + a.kh;     // input rows a band of outputs touches
  const int strips_w = (a.out_w + 3) >> 2;
  const long long nwork = (long long)a.major * bands;
  for (long long w = blockIdx.x; w < nwork; w += gridDim.x) {
    const long long mj = w / bands;
    const int band = (int)(w - mj * bands);
    const int oy0 = band * band_rows;
    const int nrows_out = min(band_rows, a.out_h - oy0);
    const int iy0 = oy0 * DOWN - a.pad_y0;                    // first input row of the band (may be negative)
    const float* src = a.in + mj * (long long)a.in_h * a.in_w;
    __syncthreads();                                          // previous band fully consumed (and sk written)
    // stage: tile[r][pad_x0 + x] = in[iy0 + r][x], zeros elsewhere
    const int nrows_in = (nrows_out - 1) * DOWN + a.kh;
    for (int e = threadIdx.x; e < nrows_in * lds_w; e += nthr) {
      const int r = e / lds_w, c = e - r * lds_w;
      const int iy = iy0 + r, ix = c - a.pad_x0;
      tile[e] = (iy >= 0 && iy < a.in_h && ix >= 0 && ix < a.in_w) ? src[(long long)iy * a.in_w + ix] : 0.f;
    }
    __syncthreads();
    float* dst = a.out + (mj * a.out_h + oy0) * (long long)a.out_w;
    for (int sidx = threadIdx.x; sidx < nrows_out * strips_w; sidx += nthr) {
      const int ry = sidx / strips_w, sx = sidx - ry * strips_w;
      const int ox = sx * 4;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      const float* trow = tile + (ry * DOWN) * lds_w + ox * DOWN;
#pragma unroll
      for (int ky = 0; ky < 4; ++ky) {
        if (ky < a.kh) {
          float x[3 * DOWN + 4];
#pragma unroll
          for (int t = 0; t < 3 * DOWN + 4; ++t) x[t] = trow[ky * lds_w + t];
#pragma unroll
          for (int kx = 0; kx < 4; ++kx) {
            const float c = sk[ky * 4 + kx];
#pragma unroll
            for (int o = 0; o < 4; ++o) v[o] = fmaf(x[o * DOWN + kx], c, v[o]);
          }
        }
      }
      float* q = dst + (long long)ry * a.out_w + ox;
#pragma unroll
      for (int o = 0; o < 4; ++o)
        if (ox + o < a.out_w) q[o] = v[o];
    }
  }
  (void)in_rows_band;
}

// 1x1 convolution with at most four input channels (the RGB input convs, discriminator.py:457-459): an outer product
// per pixel, pure streaming — y[b][o][p] = sum_c w[o][c] x[b][c][p].  One thread per (b, o, 4 pixels): the C input
// rows of an image are re-read by all O output channels from L2, the 537 MB output of the 64x64 stage is written once
// with 16-byte stores (the fp32 MFMA GEMM with K padded to 4 writes it at 2.9 TB/s).
__global__ __launch_bounds__(256) void conv1x1_smallk_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             float* __restrict__ y, int B, int C, int O, int hw4) {
  // workgroup = (256 float4 pixel groups, 4 output channels, image): no per-element index arithmetic (the flat form spent two
  // 64-bit divisions per float4: 132 us for the 268 MB output of the 64 x 64 stage, 2.0 TB/s), the C input float4 are loaded
  // once for four output planes, weights are uniform (scalar loads)
  const int p4 = blockIdx.x * 256 + threadIdx.x, o0 = blockIdx.y * 4, b = blockIdx.z;
  if (p4 >= hw4) return;
  float4 xv[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
    xv[c] = c < C ? reinterpret_cast<const float4*>(x + ((long long)b * C + c) * (long long)hw4 * 4)[p4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int o = o0 + j;
    if (o < O) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (c < C) {
          const float wv = w[o * C + c];
          acc.x = fmaf(wv, xv[c].x, acc.x); acc.y = fmaf(wv, xv[c].y, acc.y); acc.z = fmaf(wv, xv[c].z, acc.z); acc.w = fmaf(wv, xv[c].w, acc.w);
        }
      }
      reinterpret_cast<float4*>(y)[((long long)b * O + o) * hw4 + p4] = acc;
    }
  }
}

// Data gradient of the same layer: dx[b][c][p] = sum_o w[o][c] dy[b][o][p] — a reduction over the O channel planes,
// streamed once.  A workgroup owns 256 consecutive pixels of one image; its four waves each reduce a quarter of the
// channels (16-byte loads, lane = 4 pixels), and the four partial sums are added in wave order through LDS.
__global__ __launch_bounds__(256) void conv1x1_smallk_bwd_data_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                                      float* __restrict__ dx, int C, int O, int hw4) {
  __shared__ float4 red[3][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int blocks_per_img = (hw4 + 63) / 64;
  const int b = blockIdx.x / blocks_per_img, p4 = (blockIdx.x % blocks_per_img) * 64 + lane;
  const bool ok = p4 < hw4;
  float4 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int o_per = (O + 3) / 4, o0 = wave * o_per, o1 = min(O, o0 + o_per);
  const float4* src = reinterpret_cast<const float4*>(dy) + (long long)b * O * hw4 + p4;
  if (ok) {
#pragma unroll 4
    for (int o = o0; o < o1; ++o) {            // (16 loads in flight per wave measured slower: 189 us against 130)
      const float4 g = src[(long long)o * hw4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (c < C) {
          const float wv = w[o * C + c];
          acc[c].x = fmaf(wv, g.x, acc[c].x); acc[c].y = fmaf(wv, g.y, acc[c].y);
          acc[c].z = fmaf(wv, g.z, acc[c].z); acc[c].w = fmaf(wv, g.w, acc[c].w);
        }
      }
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int c = 0; c < 4; ++c) red[wave - 1][c][lane] = acc[c];
  }
  __syncthreads();
  if (wave == 0 && ok) {
    for (int c = 0; c < C; ++c) {
      float4 v = acc[c];
#pragma unroll
      for (int s2 = 0; s2 < 3; ++s2) {
        const float4 t = red[s2][c][lane];
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
      }
      reinterpret_cast<float4*>(dx)[((long long)b * C + c) * hw4 + p4] = v;
    }
  }
}

// colT[b][(c,ky,kx)][(oy,ox)] = x[b][c][oy*s+ky-p][ox*s+kx-p]   (zero outside)
// X3: write the k-major matrix as split-bf16 planes (x = hi + lo) — the B operand of the K-major / NT bf16x3 GEMMs —
// instead of fp32 (same bytes)
template <bool X3>
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ x, float* __restrict__ col,
                                                     unsigned short* __restrict__ col_hi, unsigned short* __restrict__ col_lo,
                                                     int B, int C, int H, int W, int kh, int kw, int stride, int pad, int Ho, int Wo) {
  const long long total = (long long)B * C * kh * kw * Ho * Wo;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    long long t = idx;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho); t /= Ho;
    const int kx = (int)(t % kw); t /= kw;
    const int ky = (int)(t % kh); t /= kh;
    const int c = (int)(t % C);
    const long long b = t / C;
    const int iy = oy * stride + ky - pad, ix = ox * stride + kx - pad;
    float v = 0.f;
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[((b * C + c) * H + iy) * W + ix];
    if (X3) {
      unsigned u = __float_as_uint(v);
      u += 0x7fffu + ((u >> 16) & 1u);
      const unsigned short h = (unsigned short)(u >> 16);
      unsigned l = __float_as_uint(v - __uint_as_float(((unsigned)h) << 16));
      l += 0x7fffu + ((l >> 16) & 1u);
      col_hi[idx] = h; col_lo[idx] = (unsigned short)(l >> 16);
    } else {
      col[idx] = v;
    }
  }
}

// dx[b][c][iy][ix] = sum over (ky,kx,oy,ox) hitting (iy,ix) of colT  (gather form: deterministic)
__global__ __launch_bounds__(256) void col2im_kernel(const float* __restrict__ col, float* __restrict__ dx, int B, int C,
                                                     int H, int W, int kh, int kw, int stride, int pad, int Ho, int Wo) {
  const long long total = (long long)B * C * H * W;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    long long t = idx;
    const int ix = (int)(t % W); t /= W;
    const int iy = (int)(t % H); t /= H;
    const int c = (int)(t % C);
    const long long b = t / C;
    float v = 0.f;
    for (int ky = 0; ky < kh; ++ky) {
      const int ny = iy + pad - ky;
      if (ny < 0 || (ny % stride) != 0) continue;
      const int oy = ny / stride;
      if (oy >= Ho) continue;
      for (int kx = 0; kx < kw; ++kx) {
        const int nx = ix + pad - kx;
        if (nx < 0 || (nx % stride) != 0) continue;
        const int ox = nx / stride;
        if (ox >= Wo) continue;
        v += col[((((b * C + c) * kh + ky) * kw + kx) * Ho + oy) * Wo + ox];
      }
    }
    dx[idx] = v;
  }
}

// The same gather with the filter geometry known at compile time: one workgroup per (b, c) plane, 32-bit index math,
// parity tests and divisions by constants (the generic kernel spends four 64-bit divisions and 2 kh kw runtime modulos
// per element: 839 us for the 35 M-element gradient of the 64 -> 32 stage whose bytes need 90 us).  Same taps in the same
// order (ky outer, kx inner), so the sums are those of the generic kernel bit for bit.
template <int KH, int KW, int STRIDE, int PAD>
__global__ __launch_bounds__(256) void col2im_fixed_kernel(const float* __restrict__ col, float* __restrict__ dx, int H, int W,
                                                           int Ho, int Wo) {
  // polyphase: only taps ky = (iy + PAD) % STRIDE + a * STRIDE can hit row iy.  Every candidate is LOADED (index clamped
  // to 0 when it does not exist) and selected afterwards: with the loads inside per-tap branches each element waited for
  // up to four dependent memory round trips (693 us for the 69 M-element gradient of the 64 -> 32 stage).
  constexpr int TY = (KH + STRIDE - 1) / STRIDE, TX = (KW + STRIDE - 1) / STRIDE;
  const long long plane = blockIdx.x;
  const float* cp = col + plane * (long long)(KH * KW) * Ho * Wo;
  float* dp = dx + plane * (long long)H * W;
  const unsigned n = (unsigned)(H * W), uw = (unsigned)W;
  for (unsigned e = blockIdx.y * 256 + threadIdx.x; e < n; e += gridDim.y * 256) {
    const unsigned iy = e / uw, ix = e - iy * uw;
    const int ry = ((int)iy + PAD) % STRIDE, rx = ((int)ix + PAD) % STRIDE;
    float t[TY][TX];
    bool ok[TY][TX];
#pragma unroll
    for (int a = 0; a < TY; ++a) {
      const int ky = ry + a * STRIDE, ny = (int)iy + PAD - ky, oy = ny / STRIDE;
      const bool yok = ky < KH && ny >= 0 && oy < Ho;
#pragma unroll
      for (int b = 0; b < TX; ++b) {
        const int kx = rx + b * STRIDE, nx = (int)ix + PAD - kx, ox = nx / STRIDE;
        ok[a][b] = yok && kx < KW && nx >= 0 && ox < Wo;
        const int idx = ok[a][b] ? ((ky * KW + kx) * Ho + oy) * Wo + ox : 0;
        t[a][b] = cp[idx];
      }
    }
    float v = 0.f;
#pragma unroll
    for (int a = 0; a < TY; ++a)
#pragma unroll
      for (int b = 0; b < TX; ++b) v += ok[a][b] ? t[a][b] : 0.f;
    dp[e] = v;
  }
}

inline unsigned grid_for(long long total) {
  long long b = (total + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 32768 ? 32768 : b));
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// DiffAugment (exp/cips3d/models/diffaug.py:9-85, policy 'color,translation,cutout': what Discriminator_MultiScale
// .diff_aug_img applies to every input, discriminator.py:507-508) as ONE affine operator y = A x + c and its adjoint.
//   brightness  x1 = x + (rb - 0.5)                                  (:31-33)
//   saturation  x2 = (x1 - mean_c x1) * (2 rs) + mean_c x1           (:36-39)
//   contrast    x3 = (x2 - mean_chw x2) * (rc + 0.5) + mean_chw x2   (:42-45); mean_chw x2 = mean_chw x1 exactly
//   translation x4[i][j] = x3[i + tx][j + ty], zero outside          (:48-62)
//   cutout      y = x4 * mask, mask = 0 on rows [max(ox - ch/2, 0), min(ox - ch/2 + ch - 1, H-1)] x the same in columns
// The random draws (rb, rs, rc uniform; tx, ty, ox, oy integers) are made by the host with the reference's calls, in
// its order.  Forward = per-image sum of x (for the contrast mean) + one elementwise pass; the adjoint (the backward,
// and through its own backward the R1 double-backward, train.py:387-394) has the same shape: per-image sum of the
// shifted, masked upstream gradient + one elementwise pass.  `affine` = 0 drops the brightness constant (the operator
// applied to a perturbation: backward of the adjoint).
struct DiffAugArgs {
  const float* x; float* y;
  const float *rb, *rs, *rc;                     // (B) raw uniform draws
  const long long *tx, *ty, *ox, *oy;            // (B) integer draws
  const float* sums;                             // (B) per-image sums from diffaug_sum_kernel
  int B, C, H, W, ch, cw, affine;
};

__device__ __forceinline__ bool da_masked(const DiffAugArgs& a, int b, int i, int j) {
  const int lo_i = (int)a.ox[b] - a.ch / 2, lo_j = (int)a.oy[b] - a.cw / 2;
  const int r0 = max(lo_i, 0), r1 = min(lo_i + a.ch - 1, a.H - 1);
  const int c0 = max(lo_j, 0), c1 = min(lo_j + a.cw - 1, a.W - 1);
  return i >= r0 && i <= r1 && j >= c0 && j <= c1;
}

// mode 0: sums[b] = sum of x[b]; mode 1: sums[b] = sum over the source pixels of the shifted, masked gradient
// g3[c][i'][j'] = (g * mask)[c][i' - tx][j' - ty]
// Grid (B, DA_SPLITS): block (b, s) sums slice s of image b into out[B + s * B + b]; diffaug_sum_final_kernel adds the
// slices of an image in slice order into out[b] (one workgroup per image was 223 us at 256 x 256).
constexpr int DA_SPLITS = 32;
__global__ __launch_bounds__(256) void diffaug_sum_kernel(DiffAugArgs a, float* __restrict__ out, int mode) {
  __shared__ float red[256];
  const int b = blockIdx.x;
  const int HW = a.H * a.W, n = a.C * HW;
  const int per = (n + DA_SPLITS - 1) / DA_SPLITS, e0 = blockIdx.y * per, e1 = min(n, e0 + per);
  float acc = 0.f;
  for (int e = e0 + (int)threadIdx.x; e < e1; e += 256) {
    float v = a.x[(long long)b * n + e];
    if (mode == 1) {
      const int p = e % HW, i = p / a.W, j = p - i * a.W;            // (i, j) = position in the OUTPUT of the forward
      const int si = i + (int)a.tx[b], sj = j + (int)a.ty[b];        // its source pixel
      if (da_masked(a, b, i, j) || si < 0 || si >= a.H || sj < 0 || sj >= a.W) v = 0.f;
    }
    acc += v;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s_ = 128; s_ > 0; s_ >>= 1) {
    if (threadIdx.x < s_) red[threadIdx.x] += red[threadIdx.x + s_];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[a.B + blockIdx.y * a.B + b] = red[0];
}
__global__ __launch_bounds__(256) void diffaug_sum_final_kernel(float* __restrict__ out, int B) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  float t = 0.f;
  for (int s_ = 0; s_ < DA_SPLITS; ++s_) t += out[B + s_ * B + b];
  out[b] = t;
}

__global__ __launch_bounds__(256) void diffaug_fwd_kernel(DiffAugArgs a) {
  const int HW = a.H * a.W;
  const long long total = (long long)a.B * HW;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const int b = (int)(t / HW), p = (int)(t - (long long)b * HW), i = p / a.W, j = p - i * a.W;
    const int si = i + (int)a.tx[b], sj = j + (int)a.ty[b];
    const bool zero = da_masked(a, b, i, j) || si < 0 || si >= a.H || sj < 0 || sj >= a.W;
    if (a.affine & 2) {                          // policy without the colour stage: shift and hole only, values untouched
      for (int c = 0; c < a.C; ++c)
        a.y[((long long)b * a.C + c) * HW + p] = zero ? 0.f : a.x[((long long)b * a.C + c) * HW + si * a.W + sj];
      continue;
    }
    const float off = (a.affine & 1) ? a.rb[b] - 0.5f : 0.f;
    const float s2 = a.rs[b] * 2.f, k = a.rc[b] + 0.5f;
    const float m2 = a.sums[b] / (float)(a.C * HW) + off;
    float x1[4], m1 = 0.f;
    for (int c = 0; c < a.C; ++c) {
      x1[c] = zero ? 0.f : a.x[((long long)b * a.C + c) * HW + si * a.W + sj] + off;
      m1 += x1[c];
    }
    m1 /= (float)a.C;
    for (int c = 0; c < a.C; ++c) {
      const float x2 = (x1[c] - m1) * s2 + m1;
      a.y[((long long)b * a.C + c) * HW + p] = zero ? 0.f : (x2 - m2) * k + m2;
    }
  }
}

// dx = A^T g: one thread per SOURCE pixel (i', j'): g3 = (g * mask)[i' - tx][j' - ty]; contrast^T: g2 = k g3 +
// (1 - k) / (C H W) * sum(g3); saturation^T: g1 = s g2 + (1 - s) mean_c g2; brightness^T: identity.
__global__ __launch_bounds__(256) void diffaug_adj_kernel(DiffAugArgs a) {
  const int HW = a.H * a.W;
  const long long total = (long long)a.B * HW;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const int b = (int)(t / HW), p = (int)(t - (long long)b * HW), si = p / a.W, sj = p - si * a.W;
    const int i = si - (int)a.tx[b], j = sj - (int)a.ty[b];          // where this source pixel went
    const bool none = i < 0 || i >= a.H || j < 0 || j >= a.W || da_masked(a, b, i, j);
    if (a.affine & 2) {
      for (int c = 0; c < a.C; ++c)
        a.y[((long long)b * a.C + c) * HW + p] = none ? 0.f : a.x[((long long)b * a.C + c) * HW + i * a.W + j];
      continue;
    }
    const float s2 = a.rs[b] * 2.f, k = a.rc[b] + 0.5f;
    const float spread = (1.f - k) * a.sums[b] / (float)(a.C * HW);
    float g2[4], mg = 0.f;
    for (int c = 0; c < a.C; ++c) {
      const float g3 = none ? 0.f : a.x[((long long)b * a.C + c) * HW + i * a.W + j];
      g2[c] = k * g3 + spread;
      mg += g2[c];
    }
    mg /= (float)a.C;
    for (int c = 0; c < a.C; ++c) a.y[((long long)b * a.C + c) * HW + p] = s2 * g2[c] + (1.f - s2) * mg;
  }
}

extern "C" int cips_diffaug(const float* x, float* y, const float* rb, const float* rs, const float* rc,
                            const long long* tx, const long long* ty, const long long* ox, const long long* oy,
                            float* sums, int B, int C, int H, int W, int cut_h, int cut_w, int adjoint, int affine,
                            cips_stream_t stream) {
  if (!x || !y || !rb || !rs || !rc || !tx || !ty || !ox || !oy || !sums || B <= 0 || C <= 0 || C > 4 || H <= 0 || W <= 0 ||
      cut_h < 0 || cut_w < 0)
    return (int)hipErrorInvalidValue;
  DiffAugArgs a;
  a.x = x; a.y = y; a.rb = rb; a.rs = rs; a.rc = rc; a.tx = tx; a.ty = ty; a.ox = ox; a.oy = oy; a.sums = sums;
  a.B = B; a.C = C; a.H = H; a.W = W; a.ch = cut_h; a.cw = cut_w; a.affine = affine;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(diffaug_sum_kernel, dim3(B, DA_SPLITS), dim3(256), 0, st, a, sums, adjoint ? 1 : 0);
  hipLaunchKernelGGL(diffaug_sum_final_kernel, dim3((B + 255) / 256), dim3(256), 0, st, sums, B);
  const long long total = (long long)B * H * W;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  if (adjoint) hipLaunchKernelGGL(diffaug_adj_kernel, dim3(blocks), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(diffaug_fwd_kernel, dim3(blocks), dim3(256), 0, st, a);
  return CIPS_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------------------------
// Progressive fade-in of the discriminator input (discriminator.py:524-534): F.interpolate(x, scale_factor=0.5,
// mode='bilinear') on even sizes is the 2x2 mean, h0 (w0 v00 + w1 v01) + h1 (w0 v10 + w1 v11) with all weights 0.5
// (ATen upsample_bilinear2d, align_corners False); its adjoint spreads 0.25 g.  out = a x + b y is the blend
// alpha * cur + (1 - alpha) * down (y may be NULL: out = a x, the blend's backward).
__global__ __launch_bounds__(256) void avgpool2_kernel(const float* __restrict__ x, float* __restrict__ y, long long planes, int H, int W,
                                                       int adjoint) {
  const int Ho = H / 2, Wo = W / 2;
  const long long total = planes * Ho * Wo;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const long long pl = t / (Ho * Wo);
    const int p = (int)(t - pl * (Ho * Wo)), i = p / Wo, j = p - i * Wo;
    if (!adjoint) {
      const float* q = x + pl * H * W + (2 * i) * W + 2 * j;
      const float top = __fadd_rn(0.5f * q[0], 0.5f * q[1]), bot = __fadd_rn(0.5f * q[W], 0.5f * q[W + 1]);
      y[t] = __fadd_rn(0.5f * top, 0.5f * bot);
    } else {
      const float g = 0.25f * x[t];
      float* q = y + pl * H * W + (2 * i) * W + 2 * j;
      q[0] = g; q[1] = g; q[W] = g; q[W + 1] = g;
    }
  }
}
__global__ __launch_bounds__(256) void axpby_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out, float a,
                                                    float b, long long n) {
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long long)gridDim.x * 256)
    out[t] = y ? __fadd_rn(__fmul_rn(a, x[t]), __fmul_rn(b, y[t])) : __fmul_rn(a, x[t]);
}

extern "C" int cips_avgpool2(const float* x, float* y, long long planes, int H, int W, int adjoint, cips_stream_t stream) {
  if (!x || !y || planes <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return (int)hipErrorInvalidValue;
  const long long total = planes * (H / 2) * (W / 2);
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(avgpool2_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, planes, H, W, adjoint);
  return CIPS_CHECK_LAUNCH();
}
extern "C" int cips_axpby(const float* x, const float* y, float* out, float a, float b, long long n, cips_stream_t stream) {
  if (!x || !out || n <= 0) return (int)hipErrorInvalidValue;
  const int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
  hipLaunchKernelGGL(axpby_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, out, a, b, n);
  return CIPS_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------------------------
// FID path (exp/cips3d/scripts/gen_images.py:56-60): torchvision.utils.save_image(img, normalize=True,
// value_range=(lo, hi)) turns the generator's float image into the uint8 pixels that are JPEG-encoded and fed to the
// Inception network: clamp to [lo, hi], (x - lo) * (1 / max(hi - lo, 1e-5)) (ATen's CUDA div-by-scalar is a multiply by
// the reciprocal), then mul(255).add_(0.5).clamp_(0, 255).to(uint8) in HWC order (torchvision/utils.py: make_grid
// norm_ip, save_image).  One thread per pixel; separate roundings for the multiply and the add (no fma contraction) so
// that identical float inputs give identical bytes.
__global__ __launch_bounds__(256) void image_to_u8_kernel(const float* __restrict__ x, unsigned char* __restrict__ out,
                                                          long long npix, int HW, int C, float lo, float hi, float inv) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / HW, p = i - b * HW;
    for (int c = 0; c < C; ++c) {
      float v = x[(b * C + c) * HW + p];
      v = fminf(fmaxf(v, lo), hi);
      v = __fmul_rn(__fsub_rn(v, lo), inv);
      v = __fadd_rn(__fmul_rn(v, 255.f), 0.5f);
      v = fminf(fmaxf(v, 0.f), 255.f);
      out[i * C + c] = (unsigned char)v;
    }
  }
}

extern "C" int cips_image_to_u8(const float* x, unsigned char* out, int B, int C, int H, int W, float lo, float hi,
                                cips_stream_t stream) {
  if (!x || !out || B <= 0 || C <= 0 || C > 4 || H <= 0 || W <= 0) return (int)hipErrorInvalidValue;
  const long long npix = (long long)B * H * W;
  const float d = fmaxf(hi - lo, 1e-5f);
  const int blocks = (int)((npix + 255) / 256 < 4096 ? (npix + 255) / 256 : 4096);
  hipLaunchKernelGGL(image_to_u8_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, out, npix, H * W, C, lo, hi, 1.f / d);
  return CIPS_CHECK_LAUNCH();
}

extern "C" int cips_version(void) { return 7; }     // 7: + cips_modfc_prep_bwd_batch_cores, cips_torgb_bwd_w_x3_batch_cores, cips_cores_colsum_parts (additive); 6: + cips_conv2d_x3_dgrad_s2, cips_upfirdn2d_parity, cips_lrelu_bwd_bias_finish / _nhwc, cips_conv_weight_prep_batch (additive)
extern "C" const char* cips_arch(void) { return "gfx950"; }

extern "C" int cips_fused_bias_act(const float* x, const float* bias, const float* refer, float* y,
                                   long long numel, int size_b, int step_b, int act, int grad, float alpha,
                                   float scale, cips_stream_t stream) {
  if (numel <= 0) return 0;
  if (bias && (size_b <= 0 || step_b <= 0)) return (int)hipErrorInvalidValue;
  const bool aligned = !(((uintptr_t)x | (uintptr_t)y | (uintptr_t)refer) & 15);
  // planes form: with a bias, planes of step_b elements; without one (the activation's backward: grad_output, refer = out)
  // the whole tensor is one plane
  const long long plane = bias ? step_b : numel;
  if (plane >= 64 && (plane & 3) == 0 && plane / 4 < 0x7fffffffLL && numel % plane == 0 && aligned) {
    const long long planes = numel / plane;
    const int step4 = (int)(plane / 4);
    const int by = (int)(planes < 65535 ? planes : 65535);
    const long long want = (step4 + 255) / 256, cap = 16384 / by > 1 ? 16384 / by : 1;
    const int bx = (int)(want < cap ? want : cap);
    hipLaunchKernelGGL(fused_bias_act_planes_kernel, dim3(bx, by), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(x), bias, reinterpret_cast<const float4*>(refer),
                       reinterpret_cast<float4*>(y), planes, bias ? size_b : 1, step4, act * 10 + grad, alpha, scale);
    return CIPS_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(fused_bias_act_kernel, dim3(grid_for(numel)), dim3(256), 0, (hipStream_t)stream, x, bias,
                     refer, y, numel, size_b, step_b, act * 10 + grad, alpha, scale);
  return CIPS_CHECK_LAUNCH();
}

extern "C" int cips_upfirdn2d(const float* input, const float* kernel, float* out, int major, int in_h,
                              int in_w, int minor, int kernel_h, int kernel_w, int up_x, int up_y, int down_x,
                              int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1, cips_stream_t stream) {
  if (kernel_h * kernel_w > 64 || up_x < 1 || up_y < 1 || down_x < 1 || down_y < 1) return (int)hipErrorInvalidValue;
  UpfirArgs a;
  a.in = input; a.k = kernel; a.out = out; a.major = major; a.in_h = in_h; a.in_w = in_w; a.minor = minor;
  a.kh = kernel_h; a.kw = kernel_w; a.up_x = up_x; a.up_y = up_y; a.down_x = down_x; a.down_y = down_y;
  a.pad_x0 = pad_x0; a.pad_y0 = pad_y0;
  a.out_h = (in_h * up_y + pad_y0 + pad_y1 - kernel_h) / down_y + 1;
  a.out_w = (in_w * up_x + pad_x0 + pad_x1 - kernel_w) / down_x + 1;
  long long total = (long long)major * a.out_h * a.out_w * minor;
  if (total <= 0) return 0;
  if (minor == 1 && kernel_h == 4 && kernel_w == 4 && up_x == up_y && down_x == down_y) {
    hipStream_t st = (hipStream_t)stream;
    auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
    if (up_x == 1 && down_x == 2 && pad_x0 == 1 && in_w == 2 * a.out_w && pow2(a.out_w) && a.out_w >= 2 && pad_y0 >= 0) {
      const int th = a.out_h < 48 ? 4 : 8, bh = (a.out_h + th - 1) / th, seg = a.out_w < 64 ? a.out_w : 64;
      const long long nthreads = (long long)major * a.out_w * bh;
      const unsigned grid = (unsigned)((nthreads + 255) / 256 < 131072 ? (nthreads + 255) / 256 : 131072);
      if (th == 4) hipLaunchKernelGGL(upfirdn2d_down2_pairs_kernel<4>, dim3(grid), dim3(256), 0, st, a, bh, seg);
      else hipLaunchKernelGGL(upfirdn2d_down2_pairs_kernel<8>, dim3(grid), dim3(256), 0, st, a, bh, seg);
      return CIPS_CHECK_LAUNCH();
    }
    if (up_x == 2 && down_x == 1 && pad_x0 == 2 && pad_y0 == 2 && a.out_w == 2 * in_w && pow2(in_w) && in_w >= 2 && a.out_h <= 2 * in_h) {
      const int tu = in_h < 24 ? 2 : 4, bh = (in_h + tu - 1) / tu, seg = in_w < 64 ? in_w : 64;
      const long long nthreads = (long long)major * in_w * bh;
      const unsigned grid = (unsigned)((nthreads + 255) / 256 < 131072 ? (nthreads + 255) / 256 : 131072);
      if (tu == 2) hipLaunchKernelGGL(upfirdn2d_up2_pairs_kernel<2>, dim3(grid), dim3(256), 0, st, a, bh, seg);
      else hipLaunchKernelGGL(upfirdn2d_up2_pairs_kernel<4>, dim3(grid), dim3(256), 0, st, a, bh, seg);
      return CIPS_CHECK_LAUNCH();
    }
    if (up_x == 1 && (down_x == 1 || down_x == 2)) {
      // thread tile: 1 x 16 outputs with the lanes along x (fully coalesced loads; the 4 x 4 tile of round 2 put neighbouring
      // lanes 16 B apart and moved 12.5x the output bytes through the texture path), 1 x 8 on short planes (fewer wasted rows)
      const int th = a.out_h < 48 ? 8 : 16;
      const int bw = a.out_w, bh = (a.out_h + th - 1) / th;
      const long long nthreads = (long long)major * bw * bh;
      const unsigned grid = (unsigned)((nthreads + 255) / 256 < 131072 ? (nthreads + 255) / 256 : 131072);
#define CIPS_UF(D, TW_, TH_) hipLaunchKernelGGL((upfirdn2d_direct_kernel<D, TW_, TH_, false>), dim3(grid), dim3(256), 0, st, a, bw, bh, ParityIn())
      if (down_x == 1) { if (th == 8) CIPS_UF(1, 1, 8); else CIPS_UF(1, 1, 16); }
      else { if (th == 8) CIPS_UF(2, 1, 8); else CIPS_UF(2, 1, 16); }
#undef CIPS_UF
      return CIPS_CHECK_LAUNCH();
    }
    if (up_x == 2 && down_x == 1) {
      const int bh = (a.out_h + 3) / 4;
      const long long nthreads = (long long)major * a.out_w * bh;
      const unsigned grid = (unsigned)((nthreads + 255) / 256 < 131072 ? (nthreads + 255) / 256 : 131072);
      hipLaunchKernelGGL(upfirdn2d_up2_kernel<4>, dim3(grid), dim3(256), 0, st, a, bh);
      return CIPS_CHECK_LAUNCH();
    }
  }
  if (up_x == 1 && up_y == 1 && minor == 1 && kernel_h <= 4 && kernel_w <= 4 && down_x == down_y && (down_x == 1 || down_x == 2) &&
      pad_x0 >= 0 && pad_x1 >= 0 && pad_y0 >= 0 && pad_y1 >= 0) {
    // LDS row: left padding + input + enough on the right for the last strip's reads (4 outputs past out_w at most)
    const int down = down_x;
    int lds_w = ((a.out_w + 3) / 4 * 4 - 1) * down + 4 + 3 * down + 1;
    if (lds_w < pad_x0 + in_w) lds_w = pad_x0 + in_w;
    lds_w = (lds_w + 3) & ~3;
    int band_rows = a.out_h;
    while (band_rows > 1 && ((band_rows - 1) * down + kernel_h) * lds_w > UF_MAX_LDS_FLOATS) band_rows = (band_rows + 1) / 2;
    if (((band_rows - 1) * down + kernel_h) * lds_w <= UF_MAX_LDS_FLOATS) {
      const int bands = (a.out_h + band_rows - 1) / band_rows;
      const long long nwork = (long long)major * bands;
      const int strips = band_rows * ((a.out_w + 3) / 4);                 // 4-output strips per work item
      const int nthr = strips <= 320 ? 64 : (strips <= 640 ? 128 : 256);
      const size_t lds_bytes = (size_t)(16 + ((band_rows - 1) * down + kernel_h) * lds_w) * sizeof(float);
      const long long cap = nthr == 64 ? 65536 : 16384;
      const unsigned grid = (unsigned)(nwork < cap ? nwork : cap);
      if (down == 1) hipLaunchKernelGGL(upfirdn2d_blur_kernel<1>, dim3(grid), dim3(nthr), lds_bytes, (hipStream_t)stream, a, band_rows, bands, lds_w);
      else hipLaunchKernelGGL(upfirdn2d_blur_kernel<2>, dim3(grid), dim3(nthr), lds_bytes, (hipStream_t)stream, a, band_rows, bands, lds_w);
      return CIPS_CHECK_LAUNCH();
    }
  }
  hipLaunchKernelGGL(upfirdn2d_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, a);
  return CIPS_CHECK_LAUNCH();
}

// Operand planes of the convolution WEIGHTS, all layers of a network in one launch (round 6): for every job the scaled weight
// w * scale (O, C, kh, kw) is split into bf16 hi / lo and written (a) as the forward filter bank [O][(ky, kx, c)] and (b) in ONE
// alternate form — the flipped, channel-transposed bank of the stride-1 data gradient [C][(kh-1-ky, kw-1-kx, o)] or the four
// parity banks of the stride-2 data gradient (cips_conv2d_x3_dgrad_s2).  Before, every layer built each form with its own
// multiply, permuting copy, flip and split: ~12 launches per layer, ~480 per GAN step, after every optimizer step.  Values are
// those launches' bit for bit (fp32 multiply, round-to-nearest-even splits).
// Workgroup = 32 output x 32 input channels of one job: per tap the tile is read with the lanes along c (the forward bank's
// contiguous dimension) and written back through LDS with the lanes along o (the alternate forms' contiguous dimension).
__device__ __forceinline__ unsigned short wp_f2bf(float v) {
  unsigned u = __float_as_uint(v);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
struct WPrepArgs { cips_wprep_job job[CIPS_WPREP_MAX_JOBS]; };
__global__ __launch_bounds__(256) void conv_weight_prep_kernel(WPrepArgs a) {
  __shared__ unsigned short th[32][33], tl[32][33];
  const cips_wprep_job& j = a.job[blockIdx.z];
  const int c0 = blockIdx.x * 32, o0 = blockIdx.y * 32;
  if (c0 >= j.C || o0 >= j.O) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;           // 32 x 8
  const int T = j.kh * j.kw;
  unsigned short* fh = (unsigned short*)j.fwd_hi; unsigned short* fl = (unsigned short*)j.fwd_lo;
  unsigned short* ah = (unsigned short*)j.alt_hi; unsigned short* al = (unsigned short*)j.alt_lo;
  for (int t = 0; t < T; ++t) {
    const int ky = t / j.kw, kx = t - ky * j.kw;
    for (int rr = ty; rr < 32; rr += 8) {
      const int o = o0 + rr, c = c0 + tx;
      unsigned short h = 0, l = 0;
      if (o < j.O && c < j.C) {
        float v = j.w[((long long)o * j.C + c) * T + t] * j.scale;
        asm volatile("" : "+v"(v));           // the ROUNDED product: without the barrier hipcc contracts `w * scale - hi` into one fma (lo planes 1-2 ulp off in 1.6 % of the elements)
        h = wp_f2bf(v);
        l = wp_f2bf(v - __uint_as_float(((unsigned)h) << 16));
        if (fh) { const long long q = (long long)o * T * j.C + (long long)t * j.C + c; fh[q] = h; fl[q] = l; }
      }
      th[rr][tx] = h; tl[rr][tx] = l;
    }
    __syncthreads();
    if (j.alt_kind) {
      long long base;                                               // element offset of (c = 0, o = 0) of this tap in the alternate form
      long long pitch;                                              // elements per c row
      if (j.alt_kind == 1) {
        const int tf = (j.kh - 1 - ky) * j.kw + (j.kw - 1 - kx);
        pitch = (long long)T * j.O; base = (long long)tf * j.O;
      } else {
        const int pa = ky & 1, pb = kx & 1;
        const int Ta = (j.kh - pa + 1) / 2, Tb = (j.kw - pb + 1) / 2;
        const int ty_ = Ta - 1 - (ky >> 1), tx_ = Tb - 1 - (kx >> 1);
        pitch = (long long)Ta * Tb * j.O; base = j.bank_off[2 * pa + pb] + (long long)(ty_ * Tb + tx_) * j.O;
      }
      for (int cc = ty; cc < 32; cc += 8) {
        const int c = c0 + cc, o = o0 + tx;
        if (c < j.C && o < j.O) {
          const long long q = base + (long long)c * pitch + o;
          ah[q] = th[tx][cc]; al[q] = tl[tx][cc];
        }
      }
    }
    __syncthreads();
  }
}

extern "C" int cips_conv_weight_prep_max_jobs(void) { return CIPS_WPREP_MAX_JOBS; }
extern "C" int cips_conv_weight_prep_batch(const cips_wprep_job* jobs, int njobs, cips_stream_t stream) {
  if (!jobs || njobs <= 0 || njobs > CIPS_WPREP_MAX_JOBS) return (int)hipErrorInvalidValue;
  WPrepArgs a;
  int maxO = 0, maxC = 0;
  for (int i = 0; i < njobs; ++i) {
    const cips_wprep_job& j = jobs[i];
    if (!j.w || j.O <= 0 || j.C <= 0 || j.kh <= 0 || j.kw <= 0 || j.alt_kind < 0 || j.alt_kind > 2) return (int)hipErrorInvalidValue;
    if ((j.fwd_hi == nullptr) != (j.fwd_lo == nullptr) || (j.alt_kind && (!j.alt_hi || !j.alt_lo))) return (int)hipErrorInvalidValue;
    a.job[i] = j;
    if (j.O > maxO) maxO = j.O;
    if (j.C > maxC) maxC = j.C;
  }
  dim3 grid((unsigned)((maxC + 31) / 32), (unsigned)((maxO + 31) / 32), (unsigned)njobs);
  hipLaunchKernelGGL(conv_weight_prep_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
  return CIPS_CHECK_LAUNCH();
}

// The 4 x 4 FIR (up 1, down 1) over a plane stored as the four parity blocks of cips_conv2d_x3_dgrad_s2: the transpose of
// the Blur in front of a stride-2 convolution, applied straight to that convolution's data gradient (include/cips3d_hip.h).
extern "C" int cips_upfirdn2d_parity(const float* dxp, const long long* blk_off, const float* kernel, float* out, int major,
                                     int in_h, int in_w, int pad_x0, int pad_x1, int pad_y0, int pad_y1, cips_stream_t stream) {
  if (!dxp || !blk_off || !kernel || !out || major <= 0 || in_h <= 0 || in_w <= 0 || pad_x0 < 0 || pad_y0 < 0) return (int)hipErrorInvalidValue;
  UpfirArgs a;
  a.in = dxp; a.k = kernel; a.out = out; a.major = major; a.in_h = in_h; a.in_w = in_w; a.minor = 1;
  a.kh = 4; a.kw = 4; a.up_x = a.up_y = 1; a.down_x = a.down_y = 1; a.pad_x0 = pad_x0; a.pad_y0 = pad_y0;
  a.out_h = in_h + pad_y0 + pad_y1 - 4 + 1;
  a.out_w = in_w + pad_x0 + pad_x1 - 4 + 1;
  if (a.out_h <= 0 || a.out_w <= 0) return (int)hipErrorInvalidValue;
  ParityIn pin;
  for (int pa = 0; pa < 2; ++pa)
    for (int pb = 0; pb < 2; ++pb) {
      const int hs = (in_h - pa + 1) / 2, ws = (in_w - pb + 1) / 2;
      pin.blk[2 * pa + pb] = dxp + blk_off[2 * pa + pb];
      pin.np[2 * pa + pb] = (hs * ws + 7) & ~7;
      pin.ws[pb] = ws;
    }
  const int th = a.out_h < 48 ? 8 : 16;
  const int hw2 = (a.out_w + 1) / 2, bh = (a.out_h + th - 1) / th;
  const long long nthreads = (long long)major * hw2 * bh;
  const unsigned grid = (unsigned)((nthreads + 255) / 256 < 131072 ? (nthreads + 255) / 256 : 131072);
  if (th == 8) hipLaunchKernelGGL((upfirdn2d_parity2_kernel<8>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a, hw2, bh, pin);
  else hipLaunchKernelGGL((upfirdn2d_parity2_kernel<16>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a, hw2, bh, pin);
  return CIPS_CHECK_LAUNCH();
}

// Backward of FusedLeakyReLU on a (B, C, H, W) activation with the bias gradient in the same pass (fused_act.py:26-44:
// grad_input = fused_bias_act(grad_output, empty, out, 3, 1, slope, scale); grad_bias = grad_input.sum(0, 2, 3)):
//   gin = (ref > 0 ? g : g * alpha) * scale,   part[plane][slice] = sum of gin over the slice of plane (b, c).
// The caller adds part over images and slices (B * C * slices floats) instead of re-reading the whole gradient.
__global__ __launch_bounds__(256) void lrelu_bwd_bias_kernel(const float* __restrict__ g, const float* __restrict__ ref,
                                                             float* __restrict__ gin, float* __restrict__ part, int hw, int per,
                                                             float alpha, float scale) {
  __shared__ float red[4];
  const long long base = (long long)blockIdx.x * hw;
  const int lo = blockIdx.y * per, hi = min(hw, lo + per);
  float acc = 0.f;
  if ((hw & 3) == 0 && (per & 3) == 0) {
    const float4* g4 = reinterpret_cast<const float4*>(g + base);
    const float4* r4 = reinterpret_cast<const float4*>(ref + base);
    float4* o4 = reinterpret_cast<float4*>(gin + base);
    for (int i = lo / 4 + (int)threadIdx.x; i < hi / 4; i += 256) {
      const float4 v = g4[i], r = r4[i];
      float4 o;
      o.x = (r.x > 0.f ? v.x : v.x * alpha) * scale; o.y = (r.y > 0.f ? v.y : v.y * alpha) * scale;
      o.z = (r.z > 0.f ? v.z : v.z * alpha) * scale; o.w = (r.w > 0.f ? v.w : v.w * alpha) * scale;
      o4[i] = o;
      acc += (o.x + o.y) + (o.z + o.w);
    }
  } else {
    for (int i = lo + (int)threadIdx.x; i < hi; i += 256) {
      const float v = g[base + i], r = ref[base + i];
      const float o = (r > 0.f ? v : v * alpha) * scale;
      gin[base + i] = o;
      acc += o;
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[(long long)blockIdx.x * gridDim.y + blockIdx.y] = (red[0] + red[1]) + (red[2] + red[3]);
}

// grad_bias[c] = sum over images and slices of part[b][c][s], in that fixed order (was a torch reduction per call: 60 launches of
// 13-32 us per D step for C floats each)
__global__ __launch_bounds__(256) void lrelu_bias_finish_kernel(const float* __restrict__ part, float* __restrict__ out, int B, int C, int S) {
  // one wave per channel: the B * S partials of a channel are strided over the lanes and reduced by shuffles (a thread per channel
  // walking them serially took 50 us once the NHWC form made S = pixel tiles: 2 048 dependent adds on two workgroups)
  const int lane = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= C) return;
  const int n = B * S;
  float acc = 0.f;
  for (int i = lane; i < n; i += 64) {
    const int b = i / S, s_ = i - b * S;
    acc += part[((long long)b * C + c) * S + s_];
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
  if (lane == 0) out[c] = acc;
}

// Weight gradient of the same layer: dw[o][c] = sum_{b,p} dy[b][o][p] x[b][c][p] — O*C dot products over B*HW pixels.
// As a GEMM this is one 128-row tile per image with a contraction of 65 536 (r256): four workgroups on the whole chip,
// 4.6 ms.  Here a workgroup owns one output channel and one slice of the pixel range of every image (16-byte loads, the
// C rows of x come from L2), reduces through shuffles + LDS and writes one partial row; the S partial rows are summed by
// the caller in a fixed order (deterministic, no atomics).
__global__ __launch_bounds__(256) void conv1x1_smallk_bwd_weight_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                        float* __restrict__ part, int B, int C, int O, int hw4,
                                                                        int per) {
  __shared__ float red[4][4];
  const int o = blockIdx.x, s = blockIdx.y;
  const int p_lo = s * per, p_hi = min(hw4, p_lo + per);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int b = 0; b < B; ++b) {
    const float4* g = reinterpret_cast<const float4*>(dy) + ((long long)b * O + o) * hw4;
    const float4* xs = reinterpret_cast<const float4*>(x) + (long long)b * C * hw4;
    for (int p = p_lo + (int)threadIdx.x; p < p_hi; p += 256) {
      const float4 gv = g[p];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (c < C) {
          const float4 xv = xs[(long long)c * hw4 + p];
          acc[c] = fmaf(gv.x, xv.x, fmaf(gv.y, xv.y, fmaf(gv.z, xv.z, fmaf(gv.w, xv.w, acc[c]))));
        }
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float v = acc[c];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    if (lane == 0) red[wave][c] = v;
  }
  __syncthreads();
  if (threadIdx.x < (unsigned)C) {
    const int c = threadIdx.x;
    part[((long long)s * O + o) * C + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
  }
}

extern "C" int cips_lrelu_bwd_bias_slices(int HW) {
  if (HW <= 0) return 0;
  int S = HW / 4096;                          // >= 4096 elements (16 per thread) per slice
  return S < 1 ? 1 : (S > 64 ? 64 : S);
}

extern "C" int cips_lrelu_bwd_bias(const float* grad, const float* refer, float* grad_in, float* part, long long planes, int HW,
                                   float alpha, float scale, cips_stream_t stream) {
  if (!grad || !refer || !grad_in || !part || planes <= 0 || planes > 0x7fffffffLL || HW <= 0) return (int)hipErrorInvalidValue;
  const int S = cips_lrelu_bwd_bias_slices(HW);
  int per = (HW + S - 1) / S;
  per = (per + 3) & ~3;
  hipLaunchKernelGGL(lrelu_bwd_bias_kernel, dim3((unsigned)planes, S), dim3(256), 0, (hipStream_t)stream, grad, refer, grad_in,
                     part, HW, per, alpha, scale);
  return CIPS_CHECK_LAUNCH();
}

extern "C" int cips_lrelu_bwd_bias_finish(const float* part, float* grad_bias, int B, int C, int S, cips_stream_t stream) {
  if (!part || !grad_bias || B <= 0 || C <= 0 || S <= 0) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(lrelu_bias_finish_kernel, dim3((unsigned)((C + 3) / 4)), dim3(256), 0, (hipStream_t)stream, part, grad_bias, B, C, S);
  return CIPS_CHECK_LAUNCH();
}

extern "C" int cips_conv1x1_smallk(const float* x, const float* w, float* y, int B, int C, int O, int HW,
                                   cips_stream_t stream) {
  if (B <= 0 || C <= 0 || C > 4 || O <= 0 || HW <= 0 || (HW & 3)) return (int)hipErrorInvalidValue;
  const int hw4 = HW / 4;
  if (B > 65535 || (O + 3) / 4 > 65535) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(conv1x1_smallk_kernel, dim3((unsigned)((hw4 + 255) / 256), (unsigned)((O + 3) / 4), (unsigned)B), dim3(256), 0,
                     (hipStream_t)stream, x, w, y, B, C, O, hw4);
  return CIPS_CHECK_LAUNCH();
}

extern "C" int cips_conv1x1_smallk_bwd_data(const float* dy, const float* w, float* dx, int B, int C, int O, int HW,
                                            cips_stream_t stream) {
  if (B <= 0 || C <= 0 || C > 4 || O <= 0 || HW <= 0 || (HW & 3)) return (int)hipErrorInvalidValue;
  const int hw4 = HW / 4;
  const long long blocks = (long long)B * ((hw4 + 63) / 64);
  if (blocks > 0x7fffffffLL) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(conv1x1_smallk_bwd_data_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dy, w, dx, C, O, hw4);
  return CIPS_CHECK_LAUNCH();
}

extern "C" int cips_conv1x1_smallk_bwd_weight_splits(int O, int HW) {
  if (O <= 0 || HW <= 0 || (HW & 3)) return 0;
  const int hw4 = HW / 4;
  int S = (2048 + O - 1) / O;                  // enough workgroups to fill 256 CUs a few times over
  if (S > hw4 / 256) S = hw4 / 256;            // but at least one full 256-thread sweep per image and slice
  return S < 1 ? 1 : S;
}

extern "C" int cips_conv1x1_smallk_bwd_weight(const float* dy, const float* x, float* part, int B, int C, int O, int HW,
                                              cips_stream_t stream) {
  if (!dy || !x || !part || B <= 0 || C <= 0 || C > 4 || O <= 0 || HW <= 0 || (HW & 3)) return (int)hipErrorInvalidValue;
  const int hw4 = HW / 4, S = cips_conv1x1_smallk_bwd_weight_splits(O, HW);
  const int per = (hw4 + S - 1) / S;
  hipLaunchKernelGGL(conv1x1_smallk_bwd_weight_kernel, dim3(O, S), dim3(256), 0, (hipStream_t)stream, dy, x, part, B, C, O, hw4, per);
  return CIPS_CHECK_LAUNCH();
}

extern "C" int cips_im2col(const float* x, float* col, int B, int C, int H, int W, int kh, int kw, int stride,
                           int pad, cips_stream_t stream) {
  const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
  long long total = (long long)B * C * kh * kw * Ho * Wo;
  if (total <= 0) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(im2col_kernel<false>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, col,
                     (unsigned short*)nullptr, (unsigned short*)nullptr, B, C, H, W, kh, kw, stride, pad, Ho, Wo);
  return CIPS_CHECK_LAUNCH();
}

extern "C" int cips_im2col_x3(const float* x, void* col_hi, void* col_lo, int B, int C, int H, int W, int kh, int kw,
                              int stride, int pad, cips_stream_t stream) {
  const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
  long long total = (long long)B * C * kh * kw * Ho * Wo;
  if (total <= 0 || !col_hi || !col_lo) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(im2col_kernel<true>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, (float*)nullptr,
                     (unsigned short*)col_hi, (unsigned short*)col_lo, B, C, H, W, kh, kw, stride, pad, Ho, Wo);
  return CIPS_CHECK_LAUNCH();
}

extern "C" int cips_col2im(const float* col, float* dx, int B, int C, int H, int W, int kh, int kw, int stride,
                           int pad, cips_stream_t stream) {
  const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
  long long total = (long long)B * C * H * W;
  if (total <= 0) return (int)hipErrorInvalidValue;
  const long long planes = (long long)B * C;
  if (planes <= 0x7fffffffLL && (long long)H * W < 0x40000000LL && (long long)kh * kw * Ho * Wo < 0x7fffffffLL) {
    const int per = (H * W + 255) / 256;
    const dim3 grid((unsigned)planes, (unsigned)(per < 8 ? per : 8)), blk(256);
    hipStream_t st = (hipStream_t)stream;
#define CIPS_C2I(KH, KW, S, P)                                                                                   \
    if (kh == KH && kw == KW && stride == S && pad == P) {                                                       \
      hipLaunchKernelGGL((col2im_fixed_kernel<KH, KW, S, P>), grid, blk, 0, st, col, dx, H, W, Ho, Wo);          \
      return CIPS_CHECK_LAUNCH();                                                                                \
    }
    CIPS_C2I(3, 3, 2, 0) CIPS_C2I(3, 3, 1, 1) CIPS_C2I(1, 1, 2, 0) CIPS_C2I(1, 1, 1, 0) CIPS_C2I(4, 4, 1, 0)
#undef CIPS_C2I
  }
  hipLaunchKernelGGL(col2im_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, col, dx, B, C, H, W,
                     kh, kw, stride, pad, Ho, Wo);
  return CIPS_CHECK_LAUNCH();
}
