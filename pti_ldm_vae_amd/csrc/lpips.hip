// LPIPS comparison tail: for one feature tap of the perceptual network, per sample
//     out[n] = 1/HW * sum_p sum_c w_c * (a_c / (|a_p| + 1e-10) - b_c / (|b_p| + 1e-10))^2 ,   |x_p| = sqrt(sum_c x_c^2)
// i.e. lpips' normalize_tensor on both feature maps, squared difference, the 1x1 `lin` layer (no bias) and the spatial
// mean, as the reference reaches them through monai.losses.PerceptualLoss("squeeze") (vae_scripts/train_vae.py:299,
// :395-397).  As torch ops this tail is ~12 elementwise / reduction launches per tap forward and twice that backward,
// every one a full pass over the feature maps (2.9 of the term's 6.9 ms at batch 32 x 256^2); here it is two passes
// forward (norms, then the weighted difference) and one pass backward.
//
// Layout: the feature maps are what torch's convolutions produce -- fp32 NCHW, contiguous.  A thread owns one pixel and
// a slice of C/S channels (lanes run along pixels: every load is a coalesced row piece); the S slices of a pixel are
// folded through LDS in slice order, the pixels of a workgroup by one wave in a fixed tree, and the per-workgroup
// partial sums are written with plain stores (the host adds the few rows up in order): no atomics, bitwise
// reproducible.  The forward keeps {|a_p|, |b_p|, q_p = sum_c w_c d_c a_c} per pixel so that the backward is a single
// pass:   d out[n] / d a_c = 2 / (HW * na) * (w_c d_c - a_c * q / (na * |a|)),   na = |a| + 1e-10.
// Where |a_p| = 0 (an all-zero pixel after ReLU) torch's autograd formula gives 0/0 = NaN; this kernel drops the
// second term there (the limit of the first term is what remains).
#include "pti_common.h"

namespace {

constexpr float LP_EPS = 1e-10f;

__host__ __device__ constexpr int lp_pix(int S) { return 256 / S; }

// d = x*ia - y*ib written as (x - y)*ia + y*(ia - ib): identical maps compare as exactly 0 whatever the compiler
// contracts into fmas (x*ia - y*ib came out as fma(x, ia, -round(y*ib)) ~ 1e-17), as they do in the torch formula,
// and close maps lose no bits to the cancellation of two large products.  dab = ia - ib, once per pixel.
__device__ __forceinline__ float lp_diff(float x, float ia, float y, float dab) { return (x - y) * ia + y * dab; }

template <int S>
__global__ __launch_bounds__(256) void lpips_tap_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                            const float* __restrict__ w, float* __restrict__ saved,
                                                            float* __restrict__ part, int C, int HW) {
  constexpr int PIX = lp_pix(S);
  __shared__ float red[4][256];
  const int tid = threadIdx.x, pl = tid % PIX, s = tid / PIX, n = blockIdx.y;
  const int p = blockIdx.x * PIX + pl;
  const bool ok = p < HW;
  const int cs = C / S, c0 = s * cs;
  const float* ap = a + ((size_t)n * C + c0) * HW + p;
  const float* bp = b + ((size_t)n * C + c0) * HW + p;
  float sa = 0.f, sb = 0.f;
  if (ok) {
#pragma unroll 8
    for (int c = 0; c < cs; ++c) {
      const float x = ap[(size_t)c * HW], y = bp[(size_t)c * HW];
      sa += x * x;
      sb += y * y;
    }
  }
  red[0][tid] = sa;
  red[1][tid] = sb;
  __syncthreads();
  float ta = 0.f, tb = 0.f;
#pragma unroll
  for (int k = 0; k < S; ++k) {
    ta += red[0][k * PIX + pl];
    tb += red[1][k * PIX + pl];
  }
  const float ra = sqrtf(ta), rb = sqrtf(tb);
  const float ia = 1.f / (ra + LP_EPS), ib = 1.f / (rb + LP_EPS), dab = ia - ib;
  float sd = 0.f, sq = 0.f;
  if (ok) {
#pragma unroll 8
    for (int c = 0; c < cs; ++c) {   // second pass: the rows were just read by this workgroup (L2 / MALL)
      const float x = ap[(size_t)c * HW], y = bp[(size_t)c * HW];
      const float d = lp_diff(x, ia, y, dab);
      const float wd = w[c0 + c] * d;
      sd += wd * d;
      sq += wd * x;
    }
  }
  red[2][tid] = sd;
  red[3][tid] = sq;
  __syncthreads();
  if (s == 0) {
    float td = 0.f, tq = 0.f;
#pragma unroll
    for (int k = 0; k < S; ++k) {
      td += red[2][k * PIX + pl];
      tq += red[3][k * PIX + pl];
    }
    if (ok) {
      float* sv = saved + (size_t)n * 3 * HW + p;
      sv[0] = ra;
      sv[HW] = rb;
      sv[2 * (size_t)HW] = tq;
    }
    red[0][pl] = ok ? td : 0.f;   // (every read of red[0] above happened before the second barrier)
  }
  __syncthreads();
  if (tid < 64) {
    float v = 0.f;
    for (int i = tid; i < PIX; i += 64) v += red[0][i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (tid == 0) part[(size_t)n * gridDim.x + blockIdx.x] = v;
  }
}

template <int S>
__global__ __launch_bounds__(256) void lpips_tap_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                            const float* __restrict__ w, const float* __restrict__ saved,
                                                            const float* __restrict__ gout, float* __restrict__ ga,
                                                            int C, int HW, float inv_hw) {
  constexpr int PIX = lp_pix(S);
  const int tid = threadIdx.x, pl = tid % PIX, s = tid / PIX, n = blockIdx.y;
  const int p = blockIdx.x * PIX + pl;
  if (p >= HW) return;
  const int cs = C / S, c0 = s * cs;
  const float* sv = saved + (size_t)n * 3 * HW + p;
  const float ra = sv[0], rb = sv[HW], q = sv[2 * (size_t)HW];
  const float ia = 1.f / (ra + LP_EPS), ib = 1.f / (rb + LP_EPS), dab = ia - ib;
  const float k = 2.f * gout[n] * inv_hw * ia;
  const float m = ra > 0.f ? q * ia / ra : 0.f;
  const size_t base = ((size_t)n * C + c0) * HW + p;
#pragma unroll 8
  for (int c = 0; c < cs; ++c) {
    const float x = a[base + (size_t)c * HW], y = b[base + (size_t)c * HW];
    const float d = lp_diff(x, ia, y, dab);
    ga[base + (size_t)c * HW] = k * (w[c0 + c] * d - m * x);
  }
}

// ---- the same comparison on NHWC 16-bit maps (the HIP trunk's layout: a fp16 [n][hw][c], b fp16, gradient bf16) ----
// LP lanes share a pixel (LP = the largest power of two <= 64 dividing c/8), lane l holding the 8-channel pieces
// l, l + LP, ... (<= LP_MAXP of them) in registers, so each map is read ONCE forward; channel sums are xor-butterflies
// over the LP lanes (every lane ends with the total, fixed order), the pixels of a workgroup are folded by thread 0.
constexpr int LP_MAXP = 4;
typedef _Float16 lp_f16x8 __attribute__((ext_vector_type(8)));

// Sum of the squares of one 8-channel fp16 piece.  NOT inlined on purpose: both maps go through the very same
// instructions, so identical maps get bit-identical norms (inlined, the compiler picked different fp16 -> fp32 / fma
// forms for the two maps and identical images compared as 1e-16 instead of exactly 0).
__device__ __attribute__((noinline)) float lp_sumsq8(u32x4 piece) {
  const lp_f16x8 v = __builtin_bit_cast(lp_f16x8, piece);
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float x = (float)v[e];
    s += x * x;
  }
  return s;
}

template <int LP>
__device__ __forceinline__ float lp_group_sum(float v) {
#pragma unroll
  for (int o = LP / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int LP>
__global__ __launch_bounds__(256) void lpips_tap_nhwc_fwd_kernel(const u32x4* __restrict__ a, const u32x4* __restrict__ b,
                                                                 const float* __restrict__ w, float* __restrict__ saved,
                                                                 float* __restrict__ part, int NC, int HW) {
  constexpr int PIX = 256 / LP;
  __shared__ float red[PIX];
  const int tid = threadIdx.x, l = tid % LP, pl = tid / LP, n = blockIdx.y;
  const int p = blockIdx.x * PIX + pl;
  const bool ok = p < HW;
  const int ppl = NC / LP;
  float xa[LP_MAXP][8], xb[LP_MAXP][8];
  float sa = 0.f, sb = 0.f;
#pragma unroll
  for (int j = 0; j < LP_MAXP; ++j) {
    if (j < ppl && ok) {
      const size_t o = ((size_t)n * HW + p) * NC + l + j * LP;
      const u32x4 ra_ = a[o], rb_ = b[o];
      const lp_f16x8 va = __builtin_bit_cast(lp_f16x8, ra_), vb = __builtin_bit_cast(lp_f16x8, rb_);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        xa[j][e] = (float)va[e];
        xb[j][e] = (float)vb[e];
      }
      sa += lp_sumsq8(ra_);
      sb += lp_sumsq8(rb_);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) xa[j][e] = xb[j][e] = 0.f;
    }
  }
  const float ra = sqrtf(lp_group_sum<LP>(sa)), rb = sqrtf(lp_group_sum<LP>(sb));
  const float ia = 1.f / (ra + LP_EPS), ib = 1.f / (rb + LP_EPS), dab = ia - ib;
  float sd = 0.f, sq = 0.f;
#pragma unroll
  for (int j = 0; j < LP_MAXP; ++j) {
    if (j < ppl) {
      const float* wp = w + (l + j * LP) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = lp_diff(xa[j][e], ia, xb[j][e], dab);
        const float wd = wp[e] * d;
        sd += wd * d;
        sq += wd * xa[j][e];
      }
    }
  }
  sd = lp_group_sum<LP>(sd);
  sq = lp_group_sum<LP>(sq);
  if (l == 0) {
    if (ok) {
      float* sv = saved + (size_t)n * 3 * HW + p;
      sv[0] = ra;
      sv[HW] = rb;
      sv[2 * (size_t)HW] = sq;
    }
    red[pl] = ok ? sd : 0.f;
  }
  __syncthreads();
  if (tid == 0) {
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < PIX; ++i) v += red[i];
    part[(size_t)n * gridDim.x + blockIdx.x] = v;
  }
}

template <int LP>
__global__ __launch_bounds__(256) void lpips_tap_nhwc_bwd_kernel(const u32x4* __restrict__ a, const u32x4* __restrict__ b,
                                                                 const float* __restrict__ w, const float* __restrict__ saved,
                                                                 const float* __restrict__ gout, u32x4* __restrict__ ga,
                                                                 int NC, int HW, float inv_hw) {
  constexpr int PIX = 256 / LP;
  const int tid = threadIdx.x, l = tid % LP, pl = tid / LP, n = blockIdx.y;
  const int p = blockIdx.x * PIX + pl;
  if (p >= HW) return;
  const int ppl = NC / LP;
  const float* sv = saved + (size_t)n * 3 * HW + p;
  const float ra = sv[0], rb = sv[HW], q = sv[2 * (size_t)HW];
  const float ia = 1.f / (ra + LP_EPS), ib = 1.f / (rb + LP_EPS), dab = ia - ib;
  const float k = 2.f * gout[n] * inv_hw * ia;
  const float m = ra > 0.f ? q * ia / ra : 0.f;
  for (int j = 0; j < ppl; ++j) {
    const size_t o = ((size_t)n * HW + p) * NC + l + j * LP;
    const lp_f16x8 va = __builtin_bit_cast(lp_f16x8, a[o]), vb = __builtin_bit_cast(lp_f16x8, b[o]);
    const float* wp = w + (l + j * LP) * 8;
    float g[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = (float)va[e];
      g[e] = k * (wp[e] * lp_diff(x, ia, (float)vb[e], dab) - m * x);
    }
    ga[o] = pack8(g);
  }
}

// lanes per pixel of the NHWC kernels (0: unsupported channel count)
int lp_lanes(int c) {
  if (c <= 0 || c % 8) return 0;
  const int nc = c / 8;
  for (int lp = 64; lp >= 8; lp /= 2)
    if (nc % lp == 0 && nc / lp <= LP_MAXP) return lp;
  return 0;
}

// channel slices per pixel: enough that a thread walks <= 64 channels (the small late taps would otherwise be a few
// dozen workgroups of 512-step serial loops), and S | C
int lp_slices(int c) {
  for (int s = 1; s <= 8; s *= 2)
    if (c % s == 0 && c / s <= 64) return s;
  for (int s = 8; s >= 1; s /= 2)
    if (c % s == 0) return s;
  return 1;
}

}  // namespace

extern "C" int pti_lpips_tap_blocks(int c, int hw) {
  if (c <= 0 || hw <= 0) return 0;
  const int pix = lp_pix(lp_slices(c));
  return (hw + pix - 1) / pix;
}

extern "C" int pti_lpips_tap_fwd(const float* a, const float* b, const float* w, float* saved, float* partials, int n,
                                 int c, int hw, pti_stream_t s) {
  if (!a || !b || !w || !saved || !partials) PTI_FAIL(PTI_EINVAL, "lpips_tap_fwd: null pointer");
  if (n <= 0 || c <= 0 || hw <= 0 || n > 65535) PTI_FAIL(PTI_EINVAL, "lpips_tap_fwd: bad dims n=%d c=%d hw=%d", n, c, hw);
  const int S = lp_slices(c);
  const dim3 grid((hw + lp_pix(S) - 1) / lp_pix(S), n);
  hipStream_t st = (hipStream_t)s;
  switch (S) {
    case 1: PTI_LAUNCH(lpips_tap_fwd_kernel<1>, grid, dim3(256), 0, st, a, b, w, saved, partials, c, hw); break;
    case 2: PTI_LAUNCH(lpips_tap_fwd_kernel<2>, grid, dim3(256), 0, st, a, b, w, saved, partials, c, hw); break;
    case 4: PTI_LAUNCH(lpips_tap_fwd_kernel<4>, grid, dim3(256), 0, st, a, b, w, saved, partials, c, hw); break;
    default: PTI_LAUNCH(lpips_tap_fwd_kernel<8>, grid, dim3(256), 0, st, a, b, w, saved, partials, c, hw); break;
  }
  PTI_CHECK_LAUNCH("lpips_tap_fwd");
  return PTI_OK;
}

extern "C" int pti_lpips_tap_bwd(const float* a, const float* b, const float* w, const float* saved, const float* gout,
                                 float* ga, int n, int c, int hw, pti_stream_t s) {
  if (!a || !b || !w || !saved || !gout || !ga) PTI_FAIL(PTI_EINVAL, "lpips_tap_bwd: null pointer");
  if (n <= 0 || c <= 0 || hw <= 0 || n > 65535) PTI_FAIL(PTI_EINVAL, "lpips_tap_bwd: bad dims n=%d c=%d hw=%d", n, c, hw);
  const int S = lp_slices(c);
  const dim3 grid((hw + lp_pix(S) - 1) / lp_pix(S), n);
  hipStream_t st = (hipStream_t)s;
  const float inv_hw = 1.0f / (float)hw;
  switch (S) {
    case 1: PTI_LAUNCH(lpips_tap_bwd_kernel<1>, grid, dim3(256), 0, st, a, b, w, saved, gout, ga, c, hw, inv_hw); break;
    case 2: PTI_LAUNCH(lpips_tap_bwd_kernel<2>, grid, dim3(256), 0, st, a, b, w, saved, gout, ga, c, hw, inv_hw); break;
    case 4: PTI_LAUNCH(lpips_tap_bwd_kernel<4>, grid, dim3(256), 0, st, a, b, w, saved, gout, ga, c, hw, inv_hw); break;
    default: PTI_LAUNCH(lpips_tap_bwd_kernel<8>, grid, dim3(256), 0, st, a, b, w, saved, gout, ga, c, hw, inv_hw); break;
  }
  PTI_CHECK_LAUNCH("lpips_tap_bwd");
  return PTI_OK;
}

extern "C" int pti_lpips_tap_nhwc_blocks(int c, int hw) {
  const int lp = lp_lanes(c);
  if (!lp || hw <= 0) return 0;
  return (hw + 256 / lp - 1) / (256 / lp);
}

extern "C" int pti_lpips_tap_nhwc_fwd(const void* a, const void* b, const float* w, float* saved, float* partials, int n,
                                      int c, int hw, pti_stream_t s) {
  if (!a || !b || !w || !saved || !partials) PTI_FAIL(PTI_EINVAL, "lpips_tap_nhwc_fwd: null pointer");
  const int lp = lp_lanes(c);
  if (!lp) PTI_FAIL(PTI_EUNSUPPORTED, "lpips_tap_nhwc_fwd: c=%d (need 8 * L * k, L in {8,16,32,64}, k <= 4)", c);
  if (n <= 0 || hw <= 0 || n > 65535) PTI_FAIL(PTI_EINVAL, "lpips_tap_nhwc_fwd: bad dims n=%d hw=%d", n, hw);
  const dim3 grid((hw + 256 / lp - 1) / (256 / lp), n);
  hipStream_t st = (hipStream_t)s;
  const u32x4 *pa = (const u32x4*)a, *pb = (const u32x4*)b;
  switch (lp) {
    case 8: PTI_LAUNCH(lpips_tap_nhwc_fwd_kernel<8>, grid, dim3(256), 0, st, pa, pb, w, saved, partials, c / 8, hw); break;
    case 16: PTI_LAUNCH(lpips_tap_nhwc_fwd_kernel<16>, grid, dim3(256), 0, st, pa, pb, w, saved, partials, c / 8, hw); break;
    case 32: PTI_LAUNCH(lpips_tap_nhwc_fwd_kernel<32>, grid, dim3(256), 0, st, pa, pb, w, saved, partials, c / 8, hw); break;
    default: PTI_LAUNCH(lpips_tap_nhwc_fwd_kernel<64>, grid, dim3(256), 0, st, pa, pb, w, saved, partials, c / 8, hw); break;
  }
  PTI_CHECK_LAUNCH("lpips_tap_nhwc_fwd");
  return PTI_OK;
}

extern "C" int pti_lpips_tap_nhwc_bwd(const void* a, const void* b, const float* w, const float* saved, const float* gout,
                                      void* ga, int n, int c, int hw, pti_stream_t s) {
  if (!a || !b || !w || !saved || !gout || !ga) PTI_FAIL(PTI_EINVAL, "lpips_tap_nhwc_bwd: null pointer");
  const int lp = lp_lanes(c);
  if (!lp) PTI_FAIL(PTI_EUNSUPPORTED, "lpips_tap_nhwc_bwd: c=%d", c);
  if (n <= 0 || hw <= 0 || n > 65535) PTI_FAIL(PTI_EINVAL, "lpips_tap_nhwc_bwd: bad dims n=%d hw=%d", n, hw);
  const dim3 grid((hw + 256 / lp - 1) / (256 / lp), n);
  hipStream_t st = (hipStream_t)s;
  const u32x4 *pa = (const u32x4*)a, *pb = (const u32x4*)b;
  const float inv_hw = 1.0f / (float)hw;
  switch (lp) {
    case 8: PTI_LAUNCH(lpips_tap_nhwc_bwd_kernel<8>, grid, dim3(256), 0, st, pa, pb, w, saved, gout, (u32x4*)ga, c / 8, hw, inv_hw); break;
    case 16: PTI_LAUNCH(lpips_tap_nhwc_bwd_kernel<16>, grid, dim3(256), 0, st, pa, pb, w, saved, gout, (u32x4*)ga, c / 8, hw, inv_hw); break;
    case 32: PTI_LAUNCH(lpips_tap_nhwc_bwd_kernel<32>, grid, dim3(256), 0, st, pa, pb, w, saved, gout, (u32x4*)ga, c / 8, hw, inv_hw); break;
    default: PTI_LAUNCH(lpips_tap_nhwc_bwd_kernel<64>, grid, dim3(256), 0, st, pa, pb, w, saved, gout, (u32x4*)ga, c / 8, hw, inv_hw); break;
  }
  PTI_CHECK_LAUNCH("lpips_tap_nhwc_bwd");
  return PTI_OK;
}
