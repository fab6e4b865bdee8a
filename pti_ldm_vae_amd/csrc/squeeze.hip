// Element-wise / pooling passes of the perceptual network's trunk ("HIP convs reused from K1").
// Reference: vae_scripts/train_vae.py:299 -> monai PerceptualLoss("squeeze") -> lpips.LPIPS(net="squeeze") ->
// torchvision squeezenet1_1.features: Fire modules (1x1 squeeze + ReLU, 1x1 and 3x3 expands + ReLU, concatenated) and
// MaxPool2d(3, stride 2, ceil_mode=True).  The convolutions of the Fire modules run on pti_conv2d_mfma (the two expands
// as ONE 3x3 convolution whose first half holds the 1x1 weights at the centre tap, so the concatenation is free); the
// passes here are what is left between them.  Activations: NHWC fp16 (forward), gradients: NHWC bf16; 8 channels
// (16 bytes) per thread; no atomics (the pooling backward GATHERS), bitwise reproducible.
#include "pti_common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void relu_f16_kernel(u32x4* __restrict__ x, long long n8) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  f16x8 v = __builtin_bit_cast(f16x8, x[i]);
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = v[j] > (_Float16)0 ? v[j] : (_Float16)0;
  x[i] = __builtin_bit_cast(u32x4, v);
}

// g = y > 0 ? g : 0   (g bf16, y the fp16 ReLU OUTPUT)
__global__ __launch_bounds__(256) void relu_bwd_kernel(u32x4* __restrict__ g, const u32x4* __restrict__ y, long long n8) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const f16x8 yv = __builtin_bit_cast(f16x8, y[i]);
  u32x4 gv = g[i];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    uint32_t w = gv[j];
    if (!(yv[2 * j] > (_Float16)0)) w &= 0xffff0000u;
    if (!(yv[2 * j + 1] > (_Float16)0)) w &= 0x0000ffffu;
    gv[j] = w;
  }
  g[i] = gv;
}

// g = y > 0 ? g + g2 : 0   (the tap's own gradient + the gradient arriving from the layers after it, then the mask)
__global__ __launch_bounds__(256) void relu_bwd_add_kernel(u32x4* __restrict__ g, const u32x4* __restrict__ g2,
                                                           const u32x4* __restrict__ y, long long n8) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const f16x8 yv = __builtin_bit_cast(f16x8, y[i]);
  float a[8], b[8];
  unpack8(g[i], a);
  unpack8(g2[i], b);
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = yv[j] > (_Float16)0 ? a[j] + b[j] : 0.f;
  g[i] = pack8(a);
}

// MaxPool2d(3, 2, ceil_mode=True) on NHWC fp16: thread = (output pixel, 8-channel piece); windows are clipped to the map
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const u32x4* __restrict__ x, u32x4* __restrict__ y, int H,
                                                          int W, int Ho, int Wo, int NC, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int pc = (int)(i % NC);
  long long r = i / NC;
  const int ow = (int)(r % Wo); r /= Wo;
  const int oh = (int)(r % Ho);
  const long long n = r / Ho;
  f16x8 m;
#pragma unroll
  for (int j = 0; j < 8; ++j) m[j] = -(_Float16)65504.f;
  for (int dh = 0; dh < 3; ++dh) {
    const int h = 2 * oh + dh;
    if (h >= H) break;
    for (int dw = 0; dw < 3; ++dw) {
      const int w = 2 * ow + dw;
      if (w >= W) break;
      const f16x8 v = __builtin_bit_cast(f16x8, x[((n * H + h) * W + w) * NC + pc]);
#pragma unroll
      for (int j = 0; j < 8; ++j) m[j] = v[j] > m[j] ? v[j] : m[j];
    }
  }
  y[i] = __builtin_bit_cast(u32x4, m);
}

// gx[h][w] (+)= sum over the <= 4 windows (oh, ow) that contain (h, w) of gy[oh][ow] where x[h][w] == y[oh][ow]
// (the gradient goes to the maximum; ties -- which for these post-ReLU maps happen at 0 only, where the ReLU backward
// that follows zeroes the gradient anyway -- share it).  Thread = (input pixel, 8-channel piece).
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const u32x4* __restrict__ gy, const u32x4* __restrict__ x,
                                                          const u32x4* __restrict__ y, u32x4* __restrict__ gx, int H,
                                                          int W, int Ho, int Wo, int NC, long long total, int accumulate) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int pc = (int)(i % NC);
  long long r = i / NC;
  const int w = (int)(r % W); r /= W;
  const int h = (int)(r % H);
  const long long n = r / H;
  const f16x8 xv = __builtin_bit_cast(f16x8, x[i]);
  float acc[8];
  if (accumulate) unpack8(gx[i], acc);
  else {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  }
  const int oh1 = min(h >> 1, Ho - 1), oh0 = max((h - 1) >> 1, 0);   // windows with 2*oh <= h <= 2*oh + 2
  const int ow1 = min(w >> 1, Wo - 1), ow0 = max((w - 1) >> 1, 0);
  for (int oh = oh0; oh <= oh1; ++oh)
    for (int ow = ow0; ow <= ow1; ++ow) {
      const long long o = ((n * Ho + oh) * Wo + ow) * NC + pc;
      const f16x8 yv = __builtin_bit_cast(f16x8, y[o]);
      float g[8];
      unpack8(gy[o], g);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (xv[j] == yv[j]) acc[j] += g[j];
    }
  gx[i] = pack8(acc);
}

// Layout changes at the trunk's boundary (tap 0 lives in torch's layout, NCHW fp32): 64-pixel x 64-channel tiles
// transposed through LDS so that both sides move whole 256-byte / 128-byte row pieces.
//   fwd: y[n][p][c] (fp16) = x[n][c][p] (fp32)          bwd: y[n][c][p] (fp32) += g[n][p][c] (bf16)
__global__ __launch_bounds__(256) void nchw_f32_to_nhwc_f16_kernel(const float* __restrict__ x, u32x4* __restrict__ y,
                                                                   int C, int HW) {
  __shared__ float tile[64][65];
  const int tid = threadIdx.x, n = blockIdx.z, c0 = blockIdx.y * 64, p0 = blockIdx.x * 64;
  const int lane = tid & 63, row = tid >> 6;
#pragma unroll 4
  for (int cc = row; cc < 64; cc += 4) {
    const int p = p0 + lane;
    tile[cc][lane] = p < HW ? x[((size_t)n * C + c0 + cc) * HW + p] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = tid; i < 64 * 8; i += 256) {
    const int pp = i >> 3, pc = i & 7;
    if (p0 + pp < HW) {
      f16x8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (_Float16)tile[pc * 8 + j][pp];
      y[((size_t)n * HW + p0 + pp) * (C / 8) + c0 / 8 + pc] = __builtin_bit_cast(u32x4, v);
    }
  }
}

__global__ __launch_bounds__(256) void nhwc_bf16_add_to_nchw_f32_kernel(const u32x4* __restrict__ g, float* __restrict__ y,
                                                                        int C, int HW) {
  __shared__ float tile[64][65];
  const int tid = threadIdx.x, n = blockIdx.z, c0 = blockIdx.y * 64, p0 = blockIdx.x * 64;
#pragma unroll
  for (int i = tid; i < 64 * 8; i += 256) {
    const int pp = i >> 3, pc = i & 7;
    float f[8];
    if (p0 + pp < HW) unpack8(g[((size_t)n * HW + p0 + pp) * (C / 8) + c0 / 8 + pc], f);
    else {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) tile[pc * 8 + j][pp] = f[j];
  }
  __syncthreads();
  const int lane = tid & 63, row = tid >> 6;
#pragma unroll 4
  for (int cc = row; cc < 64; cc += 4) {
    const int p = p0 + lane;
    if (p < HW) y[((size_t)n * C + c0 + cc) * HW + p] += tile[cc][lane];
  }
}

// First layer of the network for a ONE-channel image (the reference repeats the channel three times and scales each
// copy: utils/losses.py ensure_three_channels + lpips' ScalingLayer): the three input channels are the same image, so
// conv(3 -> 64, 3x3, stride 2, no padding) folds into a 1 -> 64 convolution with W'[co][tap] = sum_c W[co][c][tap] /
// scale_c and b'[co] = b[co] - sum_c shift_c / scale_c * sum_tap W[co][c][tap] (exact: no padding, every tap is always
// inside the image).  w10: fp32 [10][64] = W' tap-major, then b'.  Forward fuses the ReLU and writes tap 0 as NHWC
// fp16; backward fuses the ReLU mask and the sum over the three repeated channels: dx fp32 [n][h][w].
// Thread = (pixel, 8-channel piece): 16-byte accesses, 8 lanes per pixel.
__global__ __launch_bounds__(256) void conv1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w10,
                                                        u32x4* __restrict__ y, int H, int W, int Ho, int Wo, long long total) {
  __shared__ float wl[10 * 64];
  for (int i = threadIdx.x; i < 640; i += 256) wl[i] = w10[i];
  __syncthreads();
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int pc = (int)(i & 7);
  long long r = i >> 3;
  const int ow = (int)(r % Wo); r /= Wo;
  const int oh = (int)(r % Ho);
  const long long n = r / Ho;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = wl[9 * 64 + pc * 8 + j];
  const float* xp = x + (n * H + 2 * oh) * W + 2 * ow;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const float xv = xp[(t / 3) * W + (t % 3)];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += xv * wl[t * 64 + pc * 8 + j];
  }
  f16x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (_Float16)fmaxf(acc[j], 0.f);
  y[i] = __builtin_bit_cast(u32x4, v);
}

__global__ __launch_bounds__(256) void conv1_bwd_kernel(const u32x4* __restrict__ g, const u32x4* __restrict__ t0,
                                                        const float* __restrict__ w10, float* __restrict__ dx, int H,
                                                        int W, int Ho, int Wo, long long total) {
  __shared__ float wl[9 * 64];
  for (int i = threadIdx.x; i < 576; i += 256) wl[i] = w10[i];
  __syncthreads();
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;   // total is a multiple of 8: whole pixels per wave
  const bool ok = i < total;
  const int pc = (int)(i & 7);
  long long r = i >> 3;
  const int w = (int)(r % W); r /= W;
  const int h = (int)(r % H);
  const long long n = r / H;
  float acc = 0.f;
  if (ok) {
    for (int dh = 0; dh < 3; ++dh) {
      const int hh = h - dh;
      if (hh < 0 || (hh & 1)) continue;
      const int oh = hh >> 1;
      if (oh >= Ho) continue;
      for (int dw = 0; dw < 3; ++dw) {
        const int ww = w - dw;
        if (ww < 0 || (ww & 1)) continue;
        const int ow = ww >> 1;
        if (ow >= Wo) continue;
        const long long o = (((n * Ho + oh) * Wo + ow) << 3) + pc;
        float gv[8];
        unpack8(g[o], gv);
        const float* wr = wl + (dh * 3 + dw) * 64 + pc * 8;
        if (t0) {   // ReLU mask from the forward's output (nullptr: g is already masked)
          const f16x8 tv = __builtin_bit_cast(f16x8, t0[o]);
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (tv[j] > (_Float16)0) acc += gv[j] * wr[j];
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) acc += gv[j] * wr[j];
        }
      }
    }
  }
  acc += __shfl_xor(acc, 1, 64);
  acc += __shfl_xor(acc, 2, 64);
  acc += __shfl_xor(acc, 4, 64);
  if (ok && pc == 0) dx[i >> 3] = acc;
}

inline int pool_out(int h) {
  int ho = (h - 3 + 1) / 2 + 1;          // ceil((h - 3) / 2) + 1
  if (h < 3) ho = 1;
  if ((ho - 1) * 2 >= h) --ho;           // the last window must start inside the map (torch's rule)
  return ho;
}

}  // namespace

extern "C" int pti_relu_f16(void* x, int64_t count, pti_stream_t s) {
  if (!x || count <= 0 || count % 8) PTI_FAIL(PTI_EINVAL, "relu_f16: count %lld must be a positive multiple of 8", (long long)count);
  const long long n8 = count / 8;
  PTI_LAUNCH(relu_f16_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)s, (u32x4*)x, n8);
  PTI_CHECK_LAUNCH("relu_f16");
  return PTI_OK;
}

extern "C" int pti_relu_bwd(void* g, const void* y, int64_t count, pti_stream_t s) {
  if (!g || !y || count <= 0 || count % 8) PTI_FAIL(PTI_EINVAL, "relu_bwd: count %lld must be a positive multiple of 8", (long long)count);
  const long long n8 = count / 8;
  PTI_LAUNCH(relu_bwd_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)s, (u32x4*)g,
             (const u32x4*)y, n8);
  PTI_CHECK_LAUNCH("relu_bwd");
  return PTI_OK;
}

extern "C" int pti_relu_bwd_add(void* g, const void* g2, const void* y, int64_t count, pti_stream_t s) {
  if (!g || !g2 || !y || count <= 0 || count % 8)
    PTI_FAIL(PTI_EINVAL, "relu_bwd_add: count %lld must be a positive multiple of 8", (long long)count);
  const long long n8 = count / 8;
  PTI_LAUNCH(relu_bwd_add_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)s, (u32x4*)g,
             (const u32x4*)g2, (const u32x4*)y, n8);
  PTI_CHECK_LAUNCH("relu_bwd_add");
  return PTI_OK;
}

extern "C" int pti_maxpool3s2_out(int h) { return h > 0 ? pool_out(h) : 0; }

extern "C" int pti_maxpool3s2_fwd(const void* x, void* y, int n, int h, int w, int c, pti_stream_t s) {
  if (!x || !y || n <= 0 || h <= 0 || w <= 0 || c <= 0 || c % 8) PTI_FAIL(PTI_EINVAL, "maxpool3s2_fwd: bad args (c %% 8)");
  const int ho = pool_out(h), wo = pool_out(w), nc = c / 8;
  const long long total = (long long)n * ho * wo * nc;
  PTI_LAUNCH(maxpool_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s, (const u32x4*)x,
             (u32x4*)y, h, w, ho, wo, nc, total);
  PTI_CHECK_LAUNCH("maxpool3s2_fwd");
  return PTI_OK;
}

extern "C" int pti_maxpool3s2_bwd(const void* gy, const void* x, const void* y, void* gx, int n, int h, int w, int c,
                                  int accumulate, pti_stream_t s) {
  if (!gy || !x || !y || !gx || n <= 0 || h <= 0 || w <= 0 || c <= 0 || c % 8)
    PTI_FAIL(PTI_EINVAL, "maxpool3s2_bwd: bad args (c %% 8)");
  const int ho = pool_out(h), wo = pool_out(w), nc = c / 8;
  const long long total = (long long)n * h * w * nc;
  PTI_LAUNCH(maxpool_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s, (const u32x4*)gy,
             (const u32x4*)x, (const u32x4*)y, (u32x4*)gx, h, w, ho, wo, nc, total, accumulate);
  PTI_CHECK_LAUNCH("maxpool3s2_bwd");
  return PTI_OK;
}

extern "C" int pti_nchw_f32_to_nhwc_f16(const float* x, void* y, int n, int c, int hw, pti_stream_t s) {
  if (!x || !y || n <= 0 || c <= 0 || hw <= 0 || c % 64 || n > 65535 || c / 64 > 65535)
    PTI_FAIL(PTI_EINVAL, "nchw_f32_to_nhwc_f16: bad args (c must be a multiple of 64)");
  PTI_LAUNCH(nchw_f32_to_nhwc_f16_kernel, dim3((hw + 63) / 64, c / 64, n), dim3(256), 0, (hipStream_t)s, x, (u32x4*)y, c, hw);
  PTI_CHECK_LAUNCH("nchw_f32_to_nhwc_f16");
  return PTI_OK;
}

extern "C" int pti_nhwc_bf16_add_to_nchw_f32(const void* g, float* y, int n, int c, int hw, pti_stream_t s) {
  if (!g || !y || n <= 0 || c <= 0 || hw <= 0 || c % 64 || n > 65535 || c / 64 > 65535)
    PTI_FAIL(PTI_EINVAL, "nhwc_bf16_add_to_nchw_f32: bad args (c must be a multiple of 64)");
  PTI_LAUNCH(nhwc_bf16_add_to_nchw_f32_kernel, dim3((hw + 63) / 64, c / 64, n), dim3(256), 0, (hipStream_t)s,
             (const u32x4*)g, y, c, hw);
  PTI_CHECK_LAUNCH("nhwc_bf16_add_to_nchw_f32");
  return PTI_OK;
}

extern "C" int pti_squeeze_conv1_fwd(const float* x, const float* w10, void* y, int n, int h, int w, pti_stream_t s) {
  if (!x || !w10 || !y || n <= 0 || h < 3 || w < 3) PTI_FAIL(PTI_EINVAL, "squeeze_conv1_fwd: bad args");
  const int ho = (h - 3) / 2 + 1, wo = (w - 3) / 2 + 1;
  const long long total = (long long)n * ho * wo * 8;
  PTI_LAUNCH(conv1_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s, x, w10, (u32x4*)y, h, w,
             ho, wo, total);
  PTI_CHECK_LAUNCH("squeeze_conv1_fwd");
  return PTI_OK;
}

extern "C" int pti_squeeze_conv1_bwd(const void* g, const void* t0, const float* w10, float* dx, int n, int h, int w,
                                     pti_stream_t s) {
  if (!g || !w10 || !dx || n <= 0 || h < 3 || w < 3) PTI_FAIL(PTI_EINVAL, "squeeze_conv1_bwd: bad args");
  const int ho = (h - 3) / 2 + 1, wo = (w - 3) / 2 + 1;
  const long long total = (long long)n * h * w * 8;
  PTI_LAUNCH(conv1_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s, (const u32x4*)g,
             (const u32x4*)t0, w10, dx, h, w, ho, wo, total);
  PTI_CHECK_LAUNCH("squeeze_conv1_bwd");
  return PTI_OK;
}
