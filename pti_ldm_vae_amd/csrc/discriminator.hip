// PatchDiscriminator (adversarial branch of the VAE training step) on gfx950.
//
// Reference: vae_scripts/train_vae.py:266-279 builds MONAI's PatchDiscriminator(spatial_dims=2, num_layers_d=3,
// channels=32, in_channels=1, out_channels=1, norm="INSTANCE"): five 4x4 convolutions (pad 1; strides 2,2,2,1,1;
// channels 1 -> 32 -> 64 -> 128 -> 256 -> 1), LeakyReLU(0.2) after the first four, InstanceNorm2d (no affine, eps
// 1e-5, biased variance) in front of the activation of layers 2-4; train_vae.py:399-401,447-458 use it with
// PatchAdversarialLoss(criterion="least_squares") (MSE of LeakyReLU(0.05)(logits) against 1 / 0).
//
// Design: a 4x4 convolution is lowered to  patches (im2col, 16*Cin columns in (ky, kx, c) order, bf16)  x  a 1x1
// convolution, so the forward product, the data gradient (1x1 with the transposed weight, then the col2im gather
// below) and the weight gradient (1x1 weight gradient of patches x dy) all run on the MFMA kernels of conv_mfma.hip /
// wgrad_mfma.hip through their C-ABI; this file holds what is specific to the discriminator: the patch gather with
// InstanceNorm + LeakyReLU applied on the way in, the InstanceNorm statistics, the col2im gather fused with
// LeakyReLU' and the InstanceNorm-backward partial sums, the InstanceNorm-backward apply pass and the least-squares
// loss.  The discriminator is ~1.6 GFLOP per image and pass (the VAE step is 148): these kernels are plain
// HBM-streaming code, one 16-byte piece per lane; nothing here uses floating-point atomics (block partials + the
// fixed-order pti_gn_sums_finalize), so the adversarial step keeps the training step bitwise reproducible.
#include "pti_common.h"

namespace {

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- patches of the 1-channel fp32 image: [n][h/2][w/2][32] bf16 = 16 taps (ky*4+kx) + 16 zero columns ------------
__global__ __launch_bounds__(256) void pd_im2col_image_kernel(const float* __restrict__ img, bf16* __restrict__ P, int N,
                                                              int H, int W) {
  const int Ho = H >> 1, Wo = W >> 1;
  const int total = N * Ho * Wo;
  for (int m = blockIdx.x * 256 + threadIdx.x; m < total; m += gridDim.x * 256) {
    const int ox = m % Wo, t = m / Wo, oy = t % Ho, n = t / Ho;
    float v[16];
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
      const int iy = 2 * oy - 1 + ky;
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) {
        const int ix = 2 * ox - 1 + kx;
        v[ky * 4 + kx] = ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? img[((size_t)n * H + iy) * W + ix] : 0.f;
      }
    }
    u32x4* dst = (u32x4*)(P + (size_t)m * 32);
    dst[0] = pack8(v);
    dst[1] = pack8(v + 8);
    dst[2] = u32x4{0, 0, 0, 0};
    dst[3] = u32x4{0, 0, 0, 0};
  }
}

// ---- patches of a bf16 NHWC tensor, LeakyReLU(InstanceNorm(.)) applied on the way in -----------------------------------
struct Im2colArgs {
  const bf16* src;      // [N][H][W][C]
  const float* norm;    // [N][C][2] = {mean, rstd} or null (no normalisation)
  bf16* P;              // [N][Ho][Wo][16*C], column = (ky*4 + kx)*C + c
  int N, H, W, C, Ho, Wo, stride, act;
  float slope;
};

__global__ __launch_bounds__(256) void pd_im2col_kernel(Im2colArgs a) {
  const int NC = a.C >> 3;
  const long long total = (long long)a.N * a.Ho * a.Wo * 16 * NC;
  for (long long idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += gridDim.x * 256LL) {
    const int c8 = (int)(idx % NC);
    const long long r = idx / NC;
    const int tap = (int)(r & 15);
    const long long m = r >> 4;
    const int ox = (int)(m % a.Wo);
    const long long t = m / a.Wo;
    const int oy = (int)(t % a.Ho), n = (int)(t / a.Ho);
    const int iy = oy * a.stride - 1 + (tap >> 2), ix = ox * a.stride - 1 + (tap & 3);
    u32x4 out = {0, 0, 0, 0};   // the padding is a zero of the ACTIVATED tensor
    if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) {
      out = *(const u32x4*)(a.src + (((size_t)n * a.H + iy) * a.W + ix) * a.C + c8 * 8);
      if (a.act) {
        float f[8];
        unpack8(out, f);
        if (a.norm) {
          const float* nt = a.norm + ((size_t)n * a.C + c8 * 8) * 2;
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = (f[j] - nt[2 * j]) * nt[2 * j + 1];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = f[j] > 0.f ? f[j] : f[j] * a.slope;
        out = pack8(f);
      }
    }
    *(u32x4*)(a.P + idx * 8) = out;
  }
}

// ---- InstanceNorm statistics: {mean, rstd} per (sample, channel) over HW, biased variance ---------------------------
// grid (C/32, N), 256 threads = 64 pixel lanes x 4 channel octets; fixed-order LDS tree.
__global__ __launch_bounds__(256) void pd_in_stats_kernel(const bf16* __restrict__ y, float* __restrict__ table, int HW, int C,
                                                          float eps) {
  __shared__ float red[64][4][16];
  const int n = blockIdx.y, chunk = blockIdx.x, lc = threadIdx.x & 3, lp = threadIdx.x >> 2;
  float s[8], q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = q[j] = 0.f;
  const bf16* base = y + (size_t)n * HW * C + chunk * 32 + lc * 8;
  for (int p = lp; p < HW; p += 64) {
    float f[8];
    unpack8(*(const u32x4*)(base + (size_t)p * C), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[j] += f[j]; q[j] += f[j] * f[j]; }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { red[lp][lc][j] = s[j]; red[lp][lc][8 + j] = q[j]; }
  __syncthreads();
  for (int o = 32; o > 0; o >>= 1) {
    if (lp < o) {
#pragma unroll
      for (int j = 0; j < 16; ++j) red[lp][lc][j] += red[lp + o][lc][j];
    }
    __syncthreads();
  }
  if (lp == 0) {
    const float inv = 1.0f / (float)HW;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float mean = red[0][lc][j] * inv;
      const float var = fmaxf(red[0][lc][8 + j] * inv - mean * mean, 0.f);
      float* t = table + ((size_t)n * C + chunk * 32 + lc * 8 + j) * 2;
      t[0] = mean;
      t[1] = 1.0f / sqrtf(var + eps);
    }
  }
}

// ---- col2im gather (data gradient of the patch gather) fused with LeakyReLU' and the InstanceNorm-backward sums -----
// g[n][y][x][c] = lrelu'(xhat) * sum over the taps that read (y, x) of dP;  part[n][blk][c] += {g, g * xhat}
struct Col2imArgs {
  const bf16* dP;       // [N][Ho][Wo][16*C]
  const bf16* yprev;    // [N][H][W][C]: the conv output this layer's input was derived from (pre-norm)
  const float* norm;    // [N][C][2] or null (no norm in front of the activation: xhat = yprev, no sums)
  bf16* g;              // [N][H][W][C]
  float* part;          // [N][bps][C][2] block partials (null when norm is null)
  int N, H, W, C, Ho, Wo, stride, ppb, bps;
  float slope;
};

__global__ __launch_bounds__(256) void pd_col2im_kernel(Col2imArgs a) {
  extern __shared__ float red[];   // [ppi][NC][16]
  const int n = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
  const int NC = a.C >> 3, ppi = 256 / NC, lc = tid % NC, lp = tid / NC;
  const int HW = a.H * a.W, KC = 16 * a.C;
  float mean[8], rstd[8], s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    s1[j] = s2[j] = 0.f;
    mean[j] = 0.f;
    rstd[j] = 1.f;
  }
  if (a.norm) {
    const float* nt = a.norm + ((size_t)n * a.C + lc * 8) * 2;
#pragma unroll
    for (int j = 0; j < 8; ++j) { mean[j] = nt[2 * j]; rstd[j] = nt[2 * j + 1]; }
  }
  const int p_end = min(HW, (blk + 1) * a.ppb);
  for (int p = blk * a.ppb + lp; p < p_end; p += ppi) {
    const int y = p / a.W, x = p - y * a.W;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
      int oy = y + 1 - ky;
      if (oy < 0 || (a.stride == 2 && (oy & 1))) continue;
      if (a.stride == 2) oy >>= 1;
      if (oy >= a.Ho) continue;
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) {
        int ox = x + 1 - kx;
        if (ox < 0 || (a.stride == 2 && (ox & 1))) continue;
        if (a.stride == 2) ox >>= 1;
        if (ox >= a.Wo) continue;
        float f[8];
        unpack8(*(const u32x4*)(a.dP + (((size_t)n * a.Ho + oy) * a.Wo + ox) * KC + (ky * 4 + kx) * a.C + lc * 8), f);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += f[j];
      }
    }
    float v[8];
    const size_t off = ((size_t)n * HW + p) * a.C + lc * 8;
    unpack8(*(const u32x4*)(a.yprev + off), v);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xh = (v[j] - mean[j]) * rstd[j];
      const float gj = acc[j] * (xh > 0.f ? 1.f : a.slope);
      acc[j] = gj;
      s1[j] += gj;
      s2[j] += gj * xh;
    }
    *(u32x4*)(a.g + off) = pack8(acc);
  }
  if (!a.part) return;
  float* slot = red + (lp * NC + lc) * 16;
#pragma unroll
  for (int j = 0; j < 8; ++j) { slot[j] = s1[j]; slot[8 + j] = s2[j]; }
  __syncthreads();
  for (int o = ppi >> 1; o > 0; o >>= 1) {
    if (lp < o) {
#pragma unroll
      for (int j = 0; j < 16; ++j) slot[j] += slot[o * NC * 16 + j];
    }
    __syncthreads();
  }
  if (lp == 0) {
    float* row = a.part + (((size_t)n * a.bps + blk) * a.C + lc * 8) * 2;
#pragma unroll
    for (int j = 0; j < 8; ++j) { row[2 * j] = slot[j]; row[2 * j + 1] = slot[8 + j]; }
  }
}

// gradient w.r.t. the 1-channel image: d_img (+)= scale * sum over the (<= 4) taps of dP0 [N][H/2][W/2][32]
__global__ __launch_bounds__(256) void pd_col2im_image_kernel(const bf16* __restrict__ dP, float* __restrict__ d_img, int N, int H,
                                                              int W, float scale, int accumulate) {
  const int Ho = H >> 1, Wo = W >> 1, total = N * H * W;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int x = i % W, t = i / W, y = t % H, n = t / H;
    float acc = 0.f;
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
      const int ty = y + 1 - ky;
      if (ty < 0 || (ty & 1) || (ty >> 1) >= Ho) continue;
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) {
        const int tx = x + 1 - kx;
        if (tx < 0 || (tx & 1) || (tx >> 1) >= Wo) continue;
        acc += (float)dP[(((size_t)n * Ho + (ty >> 1)) * Wo + (tx >> 1)) * 32 + ky * 4 + kx];
      }
    }
    d_img[i] = accumulate ? d_img[i] + scale * acc : scale * acc;
  }
}

// ---- InstanceNorm backward, second pass: dy = rstd * (g - mean(g) - xhat * mean(g * xhat)) ------------------------------
__global__ __launch_bounds__(256) void pd_in_bwd_apply_kernel(const bf16* __restrict__ g, const bf16* __restrict__ y,
                                                              const float* __restrict__ norm, const float* __restrict__ sums,
                                                              bf16* __restrict__ dy, int N, int HW, int C) {
  const int NC = C >> 3;
  const long long total = (long long)N * HW * NC;
  const float inv = 1.0f / (float)HW;
  for (long long idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += gridDim.x * 256LL) {
    const int c8 = (int)(idx % NC);
    const int n = (int)(idx / ((long long)HW * NC));
    const float* nt = norm + ((size_t)n * C + c8 * 8) * 2;
    const float* st = sums + ((size_t)n * C + c8 * 8) * 2;
    float fg[8], fy[8];
    unpack8(*(const u32x4*)(g + idx * 8), fg);
    unpack8(*(const u32x4*)(y + idx * 8), fy);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float rstd = nt[2 * j + 1];
      const float xh = (fy[j] - nt[2 * j]) * rstd;
      fg[j] = rstd * (fg[j] - st[2 * j] * inv - xh * st[2 * j + 1] * inv);
    }
    *(u32x4*)(dy + idx * 8) = pack8(fg);
  }
}

// ---- final block (C -> 1 channel, 4x4, stride 1, pad 1) without the patch matrix -----------------------------------
// The last convolution has ONE output channel: lowered like the others it would write, read, and read again a
// [n][ho][wo][16*C] patch matrix (236 MB at batch 32) plus its gradient for 7 MFLOP per image.  Direct kernels
// instead: the activated input LeakyReLU(InstanceNorm(y_prev)) is rebuilt on the fly from y_prev (16 MB) wherever it is
// needed; logits and their gradient are fp32 [n][ho][wo].  w: fp32 [16][C] (tap-major), C = 32..256.
struct FinalArgs {
  const bf16* yprev;     // [N][H][W][C]
  const float* norm;     // [N][C][2] or null
  const float* w;        // [16][C]
  const float* bias;     // [1]
  float* logits;         // [N][Ho][Wo]            (forward)
  const float* dlogits;  // [N][Ho][Wo]            (backward)
  bf16* g;               // [N][H][W][C]           (data gradient, LeakyReLU' applied)
  float* part;           // data gradient: [N][bps][C][2] InstanceNorm-backward partials; weight gradient: [blocks][16*C + 8]
  int N, H, W, C, Ho, Wo, ppb, bps;
  float slope;
};

// forward: 256 threads = (256 / NC) output pixels x NC channel octets; the C-long dot products are folded over the NC
// lanes of a pixel with a fixed shuffle tree.
__global__ __launch_bounds__(256) void pd_final_fwd_kernel(FinalArgs a) {
  extern __shared__ float wsm[];   // [16][C]
  const int NC = a.C >> 3, ppi = 256 / NC, tid = threadIdx.x, lc = tid % NC, lp = tid / NC;
  for (int i = tid; i < 16 * a.C; i += 256) wsm[i] = a.w[i];
  __syncthreads();
  const int total = a.N * a.Ho * a.Wo;
  const float b0 = a.bias[0];
  for (int m0 = blockIdx.x * ppi; m0 < total; m0 += gridDim.x * ppi) {
    const int m = m0 + lp;
    float acc = 0.f;
    if (m < total) {
      const int ox = m % a.Wo, t = m / a.Wo, oy = t % a.Ho, n = t / a.Ho;
      float mean[8], rstd[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { mean[j] = 0.f; rstd[j] = 1.f; }
      if (a.norm) {
        const float* nt = a.norm + ((size_t)n * a.C + lc * 8) * 2;
#pragma unroll
        for (int j = 0; j < 8; ++j) { mean[j] = nt[2 * j]; rstd[j] = nt[2 * j + 1]; }
      }
#pragma unroll
      for (int tap = 0; tap < 16; ++tap) {
        const int iy = oy - 1 + (tap >> 2), ix = ox - 1 + (tap & 3);
        if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) {
          float f[8];
          unpack8(*(const u32x4*)(a.yprev + (((size_t)n * a.H + iy) * a.W + ix) * a.C + lc * 8), f);
          const float* wr = wsm + tap * a.C + lc * 8;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float v = (f[j] - mean[j]) * rstd[j];
            v = v > 0.f ? v : v * a.slope;
            acc += v * wr[j];
          }
        }
      }
    }
    for (int o = 1; o < NC; o <<= 1) acc += __shfl_xor(acc, o, 64);
    if (m < total && lc == 0) a.logits[m] = acc + b0;
  }
}

// data gradient w.r.t. the activated input, times LeakyReLU', + InstanceNorm-backward block partials (as pd_col2im)
__global__ __launch_bounds__(256) void pd_final_dgrad_kernel(FinalArgs a) {
  extern __shared__ float sm[];    // [16][C] weights, then [ppi][NC][16] reduction scratch
  float* wsm = sm;
  float* red = sm + 16 * a.C;
  const int n = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
  const int NC = a.C >> 3, ppi = 256 / NC, lc = tid % NC, lp = tid / NC;
  const int HW = a.H * a.W;
  for (int i = tid; i < 16 * a.C; i += 256) wsm[i] = a.w[i];
  float mean[8], rstd[8], s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = s2[j] = 0.f; mean[j] = 0.f; rstd[j] = 1.f; }
  if (a.norm) {
    const float* nt = a.norm + ((size_t)n * a.C + lc * 8) * 2;
#pragma unroll
    for (int j = 0; j < 8; ++j) { mean[j] = nt[2 * j]; rstd[j] = nt[2 * j + 1]; }
  }
  __syncthreads();
  const int p_end = min(HW, (blk + 1) * a.ppb);
  for (int p = blk * a.ppb + lp; p < p_end; p += ppi) {
    const int y = p / a.W, x = p - y * a.W;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
    for (int tap = 0; tap < 16; ++tap) {
      const int oy = y + 1 - (tap >> 2), ox = x + 1 - (tap & 3);
      if ((unsigned)oy < (unsigned)a.Ho && (unsigned)ox < (unsigned)a.Wo) {
        const float d = a.dlogits[((size_t)n * a.Ho + oy) * a.Wo + ox];
        const float* wr = wsm + tap * a.C + lc * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += d * wr[j];
      }
    }
    float v[8];
    const size_t off = ((size_t)n * HW + p) * a.C + lc * 8;
    unpack8(*(const u32x4*)(a.yprev + off), v);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xh = (v[j] - mean[j]) * rstd[j];
      const float gj = acc[j] * (xh > 0.f ? 1.f : a.slope);
      acc[j] = gj;
      s1[j] += gj;
      s2[j] += gj * xh;
    }
    *(u32x4*)(a.g + off) = pack8(acc);
  }
  if (!a.part) return;
  float* slot = red + (lp * NC + lc) * 16;
#pragma unroll
  for (int j = 0; j < 8; ++j) { slot[j] = s1[j]; slot[8 + j] = s2[j]; }
  __syncthreads();
  for (int o = ppi >> 1; o > 0; o >>= 1) {
    if (lp < o) {
#pragma unroll
      for (int j = 0; j < 16; ++j) slot[j] += slot[o * NC * 16 + j];
    }
    __syncthreads();
  }
  if (lp == 0) {
    float* row = a.part + (((size_t)n * a.bps + blk) * a.C + lc * 8) * 2;
#pragma unroll
    for (int j = 0; j < 8; ++j) { row[2 * j] = slot[j]; row[2 * j + 1] = slot[8 + j]; }
  }
}

// weight gradient, input-pixel centric: dw[tap][c] = sum over INPUT pixels q of act(q)[c] * d_logits[q + (1,1) - tap], so
// every activated pixel is loaded and normalised ONCE and feeds 16 tap accumulators (the output-centric form re-read
// each pixel 16 times through a serial, latency-bound loop: 174 us at batch 32).  A block owns FW_PIX consecutive
// input pixels of the flattened [n*h*w] range; thread = (channel octet lc, pixel lane lp), 16 x 8 accumulators; the
// pixel lanes are folded through LDS in a fixed order and the block's partial row [16][C] (+ bias sum) is stored.
constexpr int FW_PIX = 256;
__global__ __launch_bounds__(256) void pd_final_wgrad_kernel(FinalArgs a) {
  extern __shared__ float red[];   // [ppi][16][C] would be too large: folded tap by tap -> [ppi][C] per tap
  const int NC = a.C >> 3, ppi = 256 / NC, tid = threadIdx.x, lc = tid % NC, lp = tid / NC;
  const int HW = a.H * a.W, total = a.N * HW;
  const int q0 = blockIdx.x * FW_PIX, q1 = min(total, q0 + FW_PIX);
  float acc[16][8];
#pragma unroll
  for (int t = 0; t < 16; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
  for (int q = q0 + lp; q < q1; q += ppi) {
    const int n = q / HW, p = q - n * HW, y = p / a.W, x = p - y * a.W;
    float f[8];
    unpack8(*(const u32x4*)(a.yprev + (size_t)q * a.C + lc * 8), f);
    if (a.norm) {
      const float* nt = a.norm + ((size_t)n * a.C + lc * 8) * 2;
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = (f[j] - nt[2 * j]) * nt[2 * j + 1];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = f[j] > 0.f ? f[j] : f[j] * a.slope;
    const float* dn = a.dlogits + (size_t)n * a.Ho * a.Wo;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int oy = y + 1 - (t >> 2), ox = x + 1 - (t & 3);
      const float d = ((unsigned)oy < (unsigned)a.Ho && (unsigned)ox < (unsigned)a.Wo) ? dn[oy * a.Wo + ox] : 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[t][j] += d * f[j];
    }
  }
  float* row = a.part + (size_t)blockIdx.x * (16 * a.C + 8);
  for (int t = 0; t < 16; ++t) {   // fold the pixel lanes, one tap at a time (LDS: ppi * C floats = 8 KiB)
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) red[lp * a.C + lc * 8 + j] = acc[t][j];
    __syncthreads();
    for (int o = ppi >> 1; o > 0; o >>= 1) {
      if (lp < o) {
#pragma unroll
        for (int j = 0; j < 8; ++j) red[lp * a.C + lc * 8 + j] += red[(lp + o) * a.C + lc * 8 + j];
      }
      __syncthreads();
    }
    if (lp == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) row[t * a.C + lc * 8 + j] = red[lc * 8 + j];
    }
  }
  // bias gradient = sum of d_logits: block b sums the slice [b * chunk, (b+1) * chunk) of the logit map, fixed tree
  __syncthreads();
  const int M = a.N * a.Ho * a.Wo, chunk = (M + gridDim.x - 1) / gridDim.x;
  float bs = 0.f;
  for (int m = blockIdx.x * chunk + tid; m < min(M, (int)(blockIdx.x + 1) * chunk); m += 256) bs += a.dlogits[m];
  red[tid] = bs;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) row[16 * a.C] = red[0];
  if (tid > 0 && tid < 8) row[16 * a.C + tid] = 0.f;
}

// ---- PatchAdversarialLoss(least_squares): mean((lrelu_slope(logit) - target)^2) and its gradient -------------------
// logits: 16-bit, element m at m * stride.  One logit per thread; block partial sums go to loss_out[1 + block] (plain
// stores) and a one-block second launch adds them in block order: fixed-order sum, no atomics.
__global__ __launch_bounds__(256) void pd_lsgan_kernel(const void* __restrict__ logits, int f16, int stride, int M, float target,
                                                       float slope, float gscale, float* __restrict__ loss_out,
                                                       bf16* __restrict__ dY) {
  __shared__ float red[256];
  const int m = blockIdx.x * 256 + threadIdx.x;
  float acc = 0.f;
  if (m < M) {
    float v;
    if (f16 == 2) {
      v = ((const float*)logits)[(size_t)m * stride];
    } else {
      const uint16_t raw = ((const uint16_t*)logits)[(size_t)m * stride];
      v = f16 ? (float)__builtin_bit_cast(_Float16, raw) : __uint_as_float((uint32_t)raw << 16);
    }
    const float e = (v > 0.f ? v : v * slope) - target;
    acc = e * e;
    if (dY && f16 == 2) {
      ((float*)dY)[(size_t)m * stride] = gscale * e * (v > 0.f ? 1.f : slope);
    } else if (dY) {
      float d[8] = {gscale * e * (v > 0.f ? 1.f : slope), 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      u32x4* row = (u32x4*)(dY + (size_t)m * stride);
      row[0] = pack8(d);
      for (int k = 1; k < stride / 8; ++k) row[k] = u32x4{0, 0, 0, 0};
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss_out[1 + blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void pd_lsgan_finalize_kernel(float* __restrict__ loss_out, int blocks, int M) {
  __shared__ float red[256];
  float acc = 0.f;
  for (int b = threadIdx.x; b < blocks; b += 256) acc += loss_out[1 + b];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss_out[0] = red[0] / (float)M;
}

int stream_blocks(long long items) {
  long long b = (items + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));   // persistent grid-stride workgroups (tiny workgroups by the ten-thousand are dispatch-bound)
}

bool pd_channels_ok(int c) { return c >= 32 && c <= 256 && !(c & (c - 1)); }

// pixels per block of the col2im pass: a multiple of the pixels one iteration covers, ~2048 blocks per launch
void col2im_plan(int n, int hw, int c, int& ppb, int& bps) {
  const int ppi = 256 / (c / 8);
  int want = cdiv(2048, n);
  ppb = cdiv(hw, want);
  if (ppb < 4 * ppi) ppb = 4 * ppi;
  ppb = cdiv(ppb, ppi) * ppi;
  bps = cdiv(hw, ppb);
}

}  // namespace

extern "C" int pti_pd_im2col_image(const float* img, void* patches, int n, int h, int w, pti_stream_t s) {
  if (!img || !patches || n <= 0 || h <= 0 || w <= 0) PTI_FAIL(PTI_EINVAL, "pd_im2col_image: bad arguments");
  if ((h | w) & 1) PTI_FAIL(PTI_EUNSUPPORTED, "pd_im2col_image: h=%d w=%d must be even", h, w);
  PTI_LAUNCH(pd_im2col_image_kernel, dim3(stream_blocks((long long)n * (h / 2) * (w / 2))), dim3(256), 0, (hipStream_t)s, img,
             (bf16*)patches, n, h, w);
  PTI_CHECK_LAUNCH("pd_im2col_image");
  return PTI_OK;
}

extern "C" int pti_pd_im2col(const void* src, const float* norm, void* patches, int n, int h, int w, int c, int stride,
                             int act, float slope, pti_stream_t s) {
  if (!src || !patches || n <= 0 || h <= 0 || w <= 0) PTI_FAIL(PTI_EINVAL, "pd_im2col: bad arguments");
  if (!pd_channels_ok(c) || (stride != 1 && stride != 2)) PTI_FAIL(PTI_EUNSUPPORTED, "pd_im2col: c=%d stride=%d", c, stride);
  if (norm && !act) PTI_FAIL(PTI_EINVAL, "pd_im2col: a normalisation table without the activation");
  Im2colArgs a;
  a.src = (const bf16*)src; a.norm = norm; a.P = (bf16*)patches;
  a.N = n; a.H = h; a.W = w; a.C = c; a.stride = stride; a.act = act; a.slope = slope;
  a.Ho = (h + 2 - 4) / stride + 1; a.Wo = (w + 2 - 4) / stride + 1;
  if (a.Ho <= 0 || a.Wo <= 0) PTI_FAIL(PTI_EINVAL, "pd_im2col: map %dx%d too small for a 4x4 window", h, w);
  PTI_LAUNCH(pd_im2col_kernel, dim3(stream_blocks((long long)n * a.Ho * a.Wo * 16 * (c / 8))), dim3(256), 0, (hipStream_t)s, a);
  PTI_CHECK_LAUNCH("pd_im2col");
  return PTI_OK;
}

extern "C" int pti_pd_in_stats(const void* y, float* table, int n, int hw, int c, float eps, pti_stream_t s) {
  if (!y || !table || n <= 0 || hw <= 0) PTI_FAIL(PTI_EINVAL, "pd_in_stats: bad arguments");
  if (!pd_channels_ok(c)) PTI_FAIL(PTI_EUNSUPPORTED, "pd_in_stats: c=%d", c);
  PTI_LAUNCH(pd_in_stats_kernel, dim3(c / 32, n), dim3(256), 0, (hipStream_t)s, (const bf16*)y, table, hw, c, eps);
  PTI_CHECK_LAUNCH("pd_in_stats");
  return PTI_OK;
}

extern "C" int pti_pd_col2im_blocks(int n, int hw, int c) {
  if (n <= 0 || hw <= 0 || !pd_channels_ok(c)) return 0;
  int ppb, bps;
  col2im_plan(n, hw, c, ppb, bps);
  return bps;
}

extern "C" int pti_pd_col2im(const void* d_patches, const void* y_prev, const float* norm, void* g, float* partials, int n,
                             int h, int w, int c, int stride, float slope, pti_stream_t s) {
  if (!d_patches || !y_prev || !g || n <= 0 || h <= 0 || w <= 0) PTI_FAIL(PTI_EINVAL, "pd_col2im: bad arguments");
  if (!pd_channels_ok(c) || (stride != 1 && stride != 2)) PTI_FAIL(PTI_EUNSUPPORTED, "pd_col2im: c=%d stride=%d", c, stride);
  if ((norm != nullptr) != (partials != nullptr)) PTI_FAIL(PTI_EINVAL, "pd_col2im: norm and partials go together");
  Col2imArgs a;
  a.dP = (const bf16*)d_patches; a.yprev = (const bf16*)y_prev; a.norm = norm; a.g = (bf16*)g; a.part = partials;
  a.N = n; a.H = h; a.W = w; a.C = c; a.stride = stride; a.slope = slope;
  a.Ho = (h + 2 - 4) / stride + 1; a.Wo = (w + 2 - 4) / stride + 1;
  col2im_plan(n, h * w, c, a.ppb, a.bps);
  PTI_LAUNCH(pd_col2im_kernel, dim3(a.bps, n), dim3(256), 256 * 16 * sizeof(float), (hipStream_t)s, a);
  PTI_CHECK_LAUNCH("pd_col2im");
  return PTI_OK;
}

extern "C" int pti_pd_col2im_image(const void* d_patches, float* d_img, int n, int h, int w, float scale, int accumulate,
                                   pti_stream_t s) {
  if (!d_patches || !d_img || n <= 0 || h <= 0 || w <= 0 || ((h | w) & 1)) PTI_FAIL(PTI_EINVAL, "pd_col2im_image: bad arguments");
  PTI_LAUNCH(pd_col2im_image_kernel, dim3(stream_blocks((long long)n * h * w)), dim3(256), 0, (hipStream_t)s,
             (const bf16*)d_patches, d_img, n, h, w, scale, accumulate);
  PTI_CHECK_LAUNCH("pd_col2im_image");
  return PTI_OK;
}

extern "C" int pti_pd_in_bwd_apply(const void* g, const void* y, const float* norm, const float* sums, void* dy, int n, int hw,
                                   int c, pti_stream_t s) {
  if (!g || !y || !norm || !sums || !dy || n <= 0 || hw <= 0) PTI_FAIL(PTI_EINVAL, "pd_in_bwd_apply: bad arguments");
  if (!pd_channels_ok(c)) PTI_FAIL(PTI_EUNSUPPORTED, "pd_in_bwd_apply: c=%d", c);
  PTI_LAUNCH(pd_in_bwd_apply_kernel, dim3(stream_blocks((long long)n * hw * (c / 8))), dim3(256), 0, (hipStream_t)s,
             (const bf16*)g, (const bf16*)y, norm, sums, (bf16*)dy, n, hw, c);
  PTI_CHECK_LAUNCH("pd_in_bwd_apply");
  return PTI_OK;
}

extern "C" int pti_pd_lsgan_blocks(int count) { return count > 0 ? (count + 255) / 256 : 0; }

extern "C" int pti_pd_lsgan(const void* logits, int logits_f16, int stride, int count, float target, float slope,
                            float grad_scale, float* loss_out, void* d_logits, pti_stream_t s) {
  if (!logits || !loss_out || count <= 0 || stride <= 0) PTI_FAIL(PTI_EINVAL, "pd_lsgan: bad arguments");
  if (logits_f16 < 0 || logits_f16 > 2) PTI_FAIL(PTI_EINVAL, "pd_lsgan: logits format %d", logits_f16);
  if (d_logits && logits_f16 != 2 && stride % 8) PTI_FAIL(PTI_EUNSUPPORTED, "pd_lsgan: gradient rows need a stride that is a multiple of 8");
  const int blocks = pti_pd_lsgan_blocks(count);
  PTI_LAUNCH(pd_lsgan_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, logits, logits_f16, stride, count, target, slope,
             grad_scale, loss_out, (bf16*)d_logits);
  PTI_CHECK_LAUNCH("pd_lsgan");
  PTI_LAUNCH(pd_lsgan_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, loss_out, blocks, count);
  PTI_CHECK_LAUNCH("pd_lsgan_finalize");
  return PTI_OK;
}

// ---- final block, direct (see pd_final_*_kernel) ----------------------------------------------------------------------
static int final_args(FinalArgs& a, const void* y_prev, const float* norm, const float* w, int n, int h, int w_, int c, float slope) {
  if (!y_prev || !w || n <= 0 || h < 3 || w_ < 3) PTI_FAIL(PTI_EINVAL, "pd_final: bad arguments");
  if (!pd_channels_ok(c)) PTI_FAIL(PTI_EUNSUPPORTED, "pd_final: c=%d", c);
  a = FinalArgs{};
  a.yprev = (const bf16*)y_prev; a.norm = norm; a.w = w;
  a.N = n; a.H = h; a.W = w_; a.C = c; a.Ho = h - 1; a.Wo = w_ - 1; a.slope = slope;
  return PTI_OK;
}

extern "C" int pti_pd_final_fwd(const void* y_prev, const float* norm, const float* w, const float* bias, float* logits, int n,
                                int h, int w_, int c, float slope, pti_stream_t s) {
  FinalArgs a;
  if (int rc = final_args(a, y_prev, norm, w, n, h, w_, c, slope)) return rc;
  if (!bias || !logits) PTI_FAIL(PTI_EINVAL, "pd_final_fwd: null pointer");
  a.bias = bias; a.logits = logits;
  const int ppi = 256 / (c / 8);
  int blocks = (n * a.Ho * a.Wo + ppi - 1) / ppi;
  if (blocks > 768) blocks = 768;   // persistent blocks: the 16*c weight table is staged into LDS once per block
  PTI_LAUNCH(pd_final_fwd_kernel, dim3(blocks), dim3(256), 16 * c * sizeof(float), (hipStream_t)s, a);
  PTI_CHECK_LAUNCH("pd_final_fwd");
  return PTI_OK;
}

extern "C" int pti_pd_final_dgrad(const float* d_logits, const void* y_prev, const float* norm, const float* w, void* g,
                                  float* partials, int n, int h, int w_, int c, float slope, pti_stream_t s) {
  FinalArgs a;
  if (int rc = final_args(a, y_prev, norm, w, n, h, w_, c, slope)) return rc;
  if (!d_logits || !g) PTI_FAIL(PTI_EINVAL, "pd_final_dgrad: null pointer");
  if ((norm != nullptr) != (partials != nullptr)) PTI_FAIL(PTI_EINVAL, "pd_final_dgrad: norm and partials go together");
  a.dlogits = d_logits; a.g = (bf16*)g; a.part = partials;
  col2im_plan(n, h * w_, c, a.ppb, a.bps);
  PTI_LAUNCH(pd_final_dgrad_kernel, dim3(a.bps, n), dim3(256), (16 * c + 256 * 16) * sizeof(float), (hipStream_t)s, a);
  PTI_CHECK_LAUNCH("pd_final_dgrad");
  return PTI_OK;
}

extern "C" int pti_pd_final_wgrad_blocks(int n, int h, int w_) { return n > 0 && h > 1 && w_ > 1 ? (n * h * w_ + FW_PIX - 1) / FW_PIX : 0; }

extern "C" int pti_pd_final_wgrad(const float* d_logits, const void* y_prev, const float* norm, float* partials, int n, int h,
                                  int w_, int c, float slope, pti_stream_t s) {
  FinalArgs a;
  const float dummy = 0.f;
  if (int rc = final_args(a, y_prev, norm, &dummy, n, h, w_, c, slope)) return rc;
  if (!d_logits || !partials) PTI_FAIL(PTI_EINVAL, "pd_final_wgrad: null pointer");
  a.w = nullptr; a.dlogits = d_logits; a.part = partials;
  PTI_LAUNCH(pd_final_wgrad_kernel, dim3(pti_pd_final_wgrad_blocks(n, h, w_)), dim3(256), (256 / (c / 8)) * c * sizeof(float),
             (hipStream_t)s, a);
  PTI_CHECK_LAUNCH("pd_final_wgrad");
  return PTI_OK;
}
