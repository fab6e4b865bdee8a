// Image-side boundary of the MFMA path for 2..8-channel images (gfx950).
//
// The VAE's conv_in / conv_out (MONAI AutoencoderKL Encoder.blocks[0] / Decoder.blocks[-1]; reference
// src/pti_ldm_vae/models/autoencoder.py:67-79, in_channels / out_channels from the config) have one narrow side.  With
// ONE image channel -- every shipped config -- the degenerate-channel kernels of conv_direct.hip do them near their byte
// floors.  With three (BASELINE.json's "synthetic 256x256x3") those kernels cost 1.65 ms more per step (conv_out 985 us
// against 135 us, weight gradients one pass over the wide tensor PER narrow channel), so for 2..8 image channels the engine
// zero-pads the narrow side to one 32-wide MFMA tile and uses the MFMA convolution / weight-gradient kernels the ResBlocks
// use.  These two passes are the boundary with the [N,C,H,W] fp32 image tensors:
//   pti_pad_nchw_to_nhwc32:   ya / yb [n][hw][32] (16 bit)  <-  x [n][c][hw] fp32, channels >= c zero.  Two copies in one pass
//                             (fp16 operand of the forward conv, bf16 operand of the weight gradient); yb may be null.
//   pti_slice_nhwc32_to_nchw: y [n][c][hw] fp32  <-  the first c channels of x [n][hw][32] (16 bit).
// Both are plain streaming passes: 4 lanes per pixel on the padded side (1-KiB coalesced wave stores), one lane per pixel
// on the slicing side (it needs 16 of every 64 bytes: the whole tensor crosses the fabric either way).
#include "pti_common.h"

namespace {

__global__ __launch_bounds__(256) void pad_nchw_to_nhwc32_kernel(const float* __restrict__ x, u32x4* __restrict__ ya,
                                                                 u32x4* __restrict__ yb, int c, int hw, long long total,
                                                                 int a_f16, int b_f16) {
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const long long pix = idx >> 2;
    u32x4 va = {0u, 0u, 0u, 0u}, vb = va;
    if ((idx & 3) == 0) {
      const long long n = pix / hw;
      const int p = (int)(pix - n * hw);
      float f[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = j < c ? x[((size_t)n * c + j) * hw + p] : 0.f;
      va = pack8f(f, a_f16 != 0);
      vb = pack8f(f, b_f16 != 0);
    }
    ya[idx] = va;
    if (yb) yb[idx] = vb;
  }
}

__global__ __launch_bounds__(256) void slice_nhwc32_to_nchw_kernel(const u32x4* __restrict__ x, float* __restrict__ y, int c,
                                                                   int hw, long long npix, int x_f16) {
  for (long long pix = (long long)blockIdx.x * 256 + threadIdx.x; pix < npix; pix += (long long)gridDim.x * 256) {
    const long long n = pix / hw;
    const int p = (int)(pix - n * hw);
    float f[8];
    unpack8f(x[pix * 4], f, x_f16 != 0);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < c) y[((size_t)n * c + j) * hw + p] = f[j];
  }
}

}  // namespace

extern "C" int pti_pad_nchw_to_nhwc32(const float* x, void* ya, void* yb, int n, int c, int hw, int a_f16, int b_f16,
                                      pti_stream_t s) {
  if (!x || !ya || n <= 0 || hw <= 0 || c < 1 || c > 8) PTI_FAIL(PTI_EINVAL, "pad_nchw_to_nhwc32: bad args (1 <= c <= 8)");
  const long long total = (long long)n * hw * 4;
  long long blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  PTI_LAUNCH(pad_nchw_to_nhwc32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, x, (u32x4*)ya, (u32x4*)yb, c, hw,
             total, a_f16, b_f16);
  PTI_CHECK_LAUNCH("pad_nchw_to_nhwc32");
  return PTI_OK;
}

extern "C" int pti_slice_nhwc32_to_nchw(const void* x, float* y, int n, int c, int hw, int x_f16, pti_stream_t s) {
  if (!x || !y || n <= 0 || hw <= 0 || c < 1 || c > 8) PTI_FAIL(PTI_EINVAL, "slice_nhwc32_to_nchw: bad args (1 <= c <= 8)");
  const long long npix = (long long)n * hw;
  long long blocks = (npix + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  PTI_LAUNCH(slice_nhwc32_to_nchw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, (const u32x4*)x, y, c, hw, npix,
             x_f16);
  PTI_CHECK_LAUNCH("slice_nhwc32_to_nchw");
  return PTI_OK;
}
