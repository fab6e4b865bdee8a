// Direct (VALU, fp32 math) 3x3 / 1x1 stride-1 convolutions for the degenerate-channel layers of
// MONAI's Encoder/Decoder (conv_in 1->32, encoder conv_out 128->4, decoder conv_in 4->128,
// decoder conv_out 32->1; SURVEY.md §2.1 K1 "degenerate").  These have no MFMA benefit (K or N of
// the implicit GEMM is < 16) and are bandwidth/latency bound, so they are plain coalesced VALU
// kernels.  The same two kernels serve their data gradients (host passes flipped/transposed
// weights); pti_wgrad_direct is their weight/bias gradient.
//
// Tensors: "wide" side = NHWC bf16 dense (>= 32 channels, multiple of 8); "narrow" side = fp32 or
// bf16 with explicit element strides (so NCHW fp32 user tensors are read/written in place).
#include "pti_common.h"

namespace {

struct DArgs {
  const void* x;
  const float* w;     // [k*k][cin][cout]
  const float* bias;
  const stat_t* in_stats;
  const float* gamma;
  const float* beta;
  void* y;
  int N, H, W, Cin, Cout, KS;
  int prologue, groups;
  float eps, inv_cnt;
  int in_f32, out_f32, w_lds;
  int wide_f16;   // the wide NHWC 16-bit tensor (output of few-cin, input of few-cout) is fp16, else bf16
  long long is[4], os[4];  // n,h,w,c element strides of the narrow tensor(s)
};

__device__ __forceinline__ float ld_narrow(const void* p, long long idx, int f32) {
  return f32 ? ((const float*)p)[idx] : (float)((const bf16*)p)[idx];
}
__device__ __forceinline__ void st_narrow(void* p, long long idx, int f32, float v) {
  if (f32) ((float*)p)[idx] = v;
  else ((bf16*)p)[idx] = (bf16)v;
}

// ---- few input channels -> many output channels (cout % 32 == 0, output NHWC 16-bit dense) ----
// thread = (pixel, 8-channel octet of the 32-cout block blockIdx.y): FOUR consecutive lanes own one pixel's 64 bytes,
// so a wave's store instruction writes 1 KiB of consecutive addresses.  (The first version gave a thread the whole
// 32-channel row of its pixel: four 16-byte stores per lane at a 64-byte lane stride = 64 partial-line requests per
// instruction -- conv_in 1 -> 32 at 256^2, batch 32, a 134-MB write, ran at 1.3 TB/s: 103 us.)  The narrow input and
// the weights are re-read by the four lanes of a pixel (same address: one request).
__global__ __launch_bounds__(256) void direct_fewcin_kernel(DArgs a) {
  if (a.wide_f16) fp16_saturate_on();   // wave-uniform: fp16 output saturates instead of overflowing to inf
  // 32-bit index arithmetic (the host checks n*h*w*4 < 2^31): the 64-bit divisions of the first version cost several
  // hundred instructions per thread -- more than the convolution itself
  const unsigned gid = blockIdx.x * 256u + threadIdx.x;
  const int c8 = (int)(gid & 3);
  const unsigned npix = (unsigned)a.N * a.H * a.W;
  const int cb = blockIdx.y * 32 + c8 * 8;
  const int pad = (a.KS - 1) / 2;
  const bool one = a.Cin == 1 && a.KS == 3;
  // one input channel (conv_in of a grey-scale model and its mirror, the data gradient of conv_out): the 9 x 8 weights
  // of this lane live in registers, the loop is 9 loads + 72 FMAs (with the weights re-read per tap: 27 loads)
  f32x2 wr[9][4];   // pairs: the FMAs below are packed (v_pk_fma_f32, two channels per instruction)
  if (one) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const f32x4 w0 = *(const f32x4*)(a.w + (size_t)t * a.Cout + cb), w1 = *(const f32x4*)(a.w + (size_t)t * a.Cout + cb + 4);
      wr[t][0] = f32x2{w0[0], w0[1]}; wr[t][1] = f32x2{w0[2], w0[3]};
      wr[t][2] = f32x2{w1[0], w1[1]}; wr[t][3] = f32x2{w1[2], w1[3]};
    }
  }
  float b8[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) b8[c] = a.bias ? a.bias[cb + c] : 0.f;
  // persistent threads (grid-stride over the pixels): a workgroup per 64 pixels was 32768 workgroups of ~150
  // instructions each at 256^2 x 32 -- bound by the dispatch rate, not by its 134-MB write
  for (unsigned pix = gid >> 2; pix < npix; pix += gridDim.x * 64u) {
    const unsigned row = pix / (unsigned)a.W;
    const int ox = (int)(pix - row * a.W);
    const int n = (int)(row / (unsigned)a.H);
    const int oy = (int)(row - (unsigned)n * a.H);
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = b8[c];
    if (one) {
      const long long nb = n * a.is[0];
      f32x2 a2[4] = {f32x2{acc[0], acc[1]}, f32x2{acc[2], acc[3]}, f32x2{acc[4], acc[5]}, f32x2{acc[6], acc[7]}};
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int iy = oy + t / 3 - 1, ix = ox + t % 3 - 1;
        float v = 0.f;
        if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) v = ld_narrow(a.x, nb + iy * a.is[1] + ix * a.is[2], a.in_f32);
        const f32x2 vv = {v, v};
#pragma unroll
        for (int c = 0; c < 4; ++c) a2[c] += vv * wr[t][c];
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) { acc[2 * c] = a2[c][0]; acc[2 * c + 1] = a2[c][1]; }
    } else {
      for (int kh = 0; kh < a.KS; ++kh) {
        const int iy = oy + kh - pad;
        if (iy < 0 || iy >= a.H) continue;
        for (int kw = 0; kw < a.KS; ++kw) {
          const int ix = ox + kw - pad;
          if (ix < 0 || ix >= a.W) continue;
          const long long ibase = n * a.is[0] + iy * a.is[1] + ix * a.is[2];
          const float* wt = a.w + (size_t)((kh * a.KS + kw) * a.Cin) * a.Cout + cb;
          for (int ci = 0; ci < a.Cin; ++ci) {
            const float v = ld_narrow(a.x, ibase + ci * a.is[3], a.in_f32);
            const f32x4 w0 = *(const f32x4*)(wt + (size_t)ci * a.Cout), w1 = *(const f32x4*)(wt + (size_t)ci * a.Cout + 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) { acc[c] += v * w0[c]; acc[4 + c] += v * w1[c]; }
          }
        }
      }
    }
    *(u32x4*)((bf16*)a.y + (size_t)pix * a.Cout + cb) = pack8f(acc, a.wide_f16);
  }
}

// ---- many input channels (NHWC 16-bit dense, cin % 8 == 0, cin/8 | 64) -> few output channels ----
// Centre-based: out[p][co] = sum_t partial_t[p + off_t][co] with partial_t[q][co] = sum_ci act(x)[q][ci] * w[t][ci][co].
// A block owns a TH x 16 tile of output pixels.  Phase 1: every pixel q of the tile + 1-pixel halo is loaded ONCE
// (16-byte pieces, NC = cin/8 consecutive lanes per pixel), the GroupNorm prologue is applied ONCE, the 9*Cout
// partial dot products are reduced over the NC lanes with a halving shuffle tree and stored to an LDS table
// P[q][t][co].  Phase 2: each output gathers its 9 partials from LDS.  (The first version gathered the 9 neighbours
// per output from global memory and re-applied the prologue to each: 0.5 TB/s at 256^2; its LDS weight table was
// also read at a 128-byte lane stride = 16-way bank conflicts: 138 us for an 8 MB map.)
template <int MAXCO>
struct FoCfg {
  static constexpr int TW = 16, TH = MAXCO == 1 ? 32 : (MAXCO <= 4 ? 8 : 4);   // 1 channel: 32x16 tile (halo overhead 1.2x instead of 1.41x, a quarter of the barriers per output: 170 -> 124 us)
  static constexpr int HW = TW + 2, HH = TH + 2, NP = HH * HW;
  static constexpr int NV = 9 * MAXCO;            // partial sums per centre pixel
};

template <int MAXCO>
__global__ __launch_bounds__(256) void direct_fewcout_kernel(DArgs a, int tiles_x, int tiles_y) {
  using C = FoCfg<MAXCO>;
  extern __shared__ float fsm[];
  float* P = fsm;                                  // [NP][NV]
  float* wsm = fsm + C::NP * C::NV;                // MAXCO > 1: [9][NC][8*Cout + 4] (padded against bank conflicts)
  const int NC = a.Cin / 8;
  const int tid = threadIdx.x, lc = tid % NC;
  const int pad = (a.KS - 1) / 2, ntap = a.KS * a.KS;
  const int ntiles = a.N * tiles_x * tiles_y;
  const int wpitch = 8 * a.Cout + 4;
  const bf16* X = (const bf16*)a.x;
  float wreg[MAXCO == 1 ? 9 : 1][8];
  if constexpr (MAXCO == 1) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) wreg[t][j] = (t < ntap) ? a.w[(size_t)(t * a.Cin + lc * 8 + j)] : 0.f;
  } else if (a.w_lds) {
    const int per_tap = a.Cin * a.Cout;
    for (int i = tid; i < ntap * per_tap; i += 256) {
      const int t = i / per_tap, r = i - t * per_tap;         // r = ci * Cout + co
      const int g = r / (8 * a.Cout), q = r - g * 8 * a.Cout;
      wsm[(t * NC + g) * wpitch + q] = a.w[i];
    }
  }
  float sc[8], sh[8];
  int cur_n = -1;
  if constexpr (MAXCO > 1) __syncthreads();
  // persistent blocks: the per-block set-up (weights to registers / LDS) is paid once, not once per 128 outputs
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
  const int tx = tile % tiles_x;
  const int tr = tile / tiles_x;
  const int n = tr / tiles_y;
  const int oy0 = (tr - n * tiles_y) * C::TH, ox0 = tx * C::TW;
  if (a.prologue && n != cur_n) {
    cur_n = n;
    const int cpg = a.Cin / a.groups;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ch = lc * 8 + j, g = ch / cpg;
      const float sum = stat_f(a.in_stats, (n * a.groups + g) * 2), sq = stat_f(a.in_stats, (n * a.groups + g) * 2 + 1);
      const float mean = sum * a.inv_cnt;
      const float rstd = rsqrtf(fmaxf(sq * a.inv_cnt - mean * mean, 0.f) + a.eps);
      sc[j] = rstd * a.gamma[ch];
      sh[j] = a.beta[ch] - mean * sc[j];
    }
  }

  // ---- phase 1: partial dot products of every centre pixel of the halo'd tile ----
  const int ppi = 256 / NC;                         // pixels per iteration
  // the halo pixels are fetched LB iterations at a time: one global round trip per batch instead of one per
  // iteration (32 channels: the whole 180-pixel halo is 3 iterations = one batch; the loop used to expose three
  // serial ~2 us round trips per 128 outputs and ran 8x off its byte roofline)
  constexpr int LB = 4;
  for (int qb = 0; qb < C::NP; qb += LB * ppi) {
  u32x4 rawb[LB];
#pragma unroll
  for (int b = 0; b < LB; ++b) {
    const int q = qb + b * ppi + tid / NC;
    const int hy = q / C::HW, hx = q - hy * C::HW;
    const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
    rawb[b] = u32x4{0u, 0u, 0u, 0u};
    if (q < C::NP && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
      rawb[b] = *(const u32x4*)(X + ((size_t)(n * a.H + iy) * a.W + ix) * a.Cin + lc * 8);
  }
#pragma unroll
  for (int b = 0; b < LB; ++b) {
    const int q0 = qb + b * ppi;
    if (q0 >= C::NP) break;
    const int q = q0 + tid / NC;
    const int hy = q / C::HW, hx = q - hy * C::HW;
    const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
    const bool ok = q < C::NP && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = 0.f;
    if (ok) {
      unpack8f(rawb[b], f, a.wide_f16);
      if (a.prologue) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float v = f[j] * sc[j] + sh[j];
          if (a.prologue == PTI_PRO_GN_SILU) v = silu_f(v);
          f[j] = v;
        }
      }
    }
    // v[t * MAXCO + co]; padded to a multiple of 16 so that the halving tree below is uniform for every NC <= 16
    constexpr int NVP = (C::NV + 15) / 16 * 16;
    float v[NVP];
#pragma unroll
    for (int i = 0; i < NVP; ++i) v[i] = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      if constexpr (MAXCO == 1) {
        float s_ = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s_ += f[j] * wreg[t][j];
        v[t] = s_;
      } else {
        if (t < ntap) {
          const float* wt = a.w_lds ? wsm + (t * NC + lc) * wpitch : a.w + (size_t)(t * a.Cin + lc * 8) * a.Cout;
#pragma unroll
          for (int c = 0; c < MAXCO; ++c) {
            if (c < a.Cout) {
              float s_ = 0.f;
#pragma unroll
              for (int j = 0; j < 8; ++j) s_ += f[j] * wt[j * a.Cout + c];
              v[t * MAXCO + c] = s_;
            }
          }
        }
      }
    }
    // reduce over the NC (>= 4) lanes of the pixel: two halving levels (a lane keeps one half of its values and
    // trades the other with its partner), then plain butterflies on the remaining NVP/4 values
    {
      const bool h1 = (lc & 1) != 0, h2 = (lc & 2) != 0;
#pragma unroll
      for (int i = 0; i < NVP / 2; ++i) {
        const float send = h1 ? v[i] : v[i + NVP / 2];
        const float keep = h1 ? v[i + NVP / 2] : v[i];
        v[i] = keep + __shfl_xor(send, 1, 64);
      }
#pragma unroll
      for (int i = 0; i < NVP / 4; ++i) {
        const float send = h2 ? v[i] : v[i + NVP / 4];
        const float keep = h2 ? v[i + NVP / 4] : v[i];
        v[i] = keep + __shfl_xor(send, 2, 64);
      }
#pragma unroll
      for (int o = 4; o < 64; o <<= 1) {
        if (o < NC) {
#pragma unroll
          for (int i = 0; i < NVP / 4; ++i) v[i] += __shfl_xor(v[i], o, 64);
        }
      }
      // lane lc (< 4) now holds values [first, first + NVP/4) of its pixel, summed over all channels
      const int first = (h1 ? NVP / 2 : 0) + (h2 ? NVP / 4 : 0);
      if (lc < 4 && q < C::NP) {
#pragma unroll
        for (int i = 0; i < NVP / 4; ++i)
          if (first + i < C::NV) P[q * C::NV + first + i] = v[i];
      }
    }
  }
  }
  __syncthreads();
  // ---- phase 2: gather ----
  for (int e = tid; e < C::TH * C::TW * MAXCO; e += 256) {
    const int co = e % MAXCO, pi = e / MAXCO;
    const int py = pi / C::TW, px = pi - py * C::TW;
    const int oy = oy0 + py, ox = ox0 + px;
    if (co < a.Cout && oy < a.H && ox < a.W) {
      float s_ = a.bias ? a.bias[co] : 0.f;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
          if (kh < a.KS && kw < a.KS)
            s_ += P[((py + 1 + kh - pad) * C::HW + px + 1 + kw - pad) * C::NV + (kh * a.KS + kw) * MAXCO + co];
      st_narrow(a.y, n * a.os[0] + oy * a.os[1] + ox * a.os[2] + co * a.os[3], a.out_f32, s_);
    }
  }
  __syncthreads();   // P is rewritten by the next tile
  }
}

// ---- weight / bias gradient of the direct convolutions -------------------------------------
// out[tap][cw] (+)= sum_p narrow[p] * T(wide)[p + sgn*(tap offset)][cw]   for one narrow channel k
//   FEWCOUT layer (wide = input x with prologue, narrow = dY[..,k=co]), sgn=+1: dW[co][ci][tap]
//   FEWCIN  layer (wide = dY,               narrow = x[..,k=ci]),      sgn=-1: dW[co][ci][tap]
// Written with atomics into fp32 gradients laid out OIHW (strides given); grid.y = narrow channel.
struct WGArgs {
  const bf16* wide;   // [N,H,W,CW] dense
  const void* narrow; // strided
  float* dw;          // fp32 OIHW gradient
  float* dbias_wide;  // optional [CW]: column sums of wide   (FEWCIN layer bias grad)
  float* dbias_narrow;// optional [narrow ch]: sum of narrow   (FEWCOUT layer bias grad)
  float* part;        // workspace: [cn][blocks][80*NC + 1] per-block partial sums (plain stores)
  const stat_t* in_stats; const float* gamma; const float* beta;
  int N, H, W, CW, KS, sgn;
  int prologue, groups; float eps, inv_cnt;
  int narrow_f32; long long ns[4];
  int wide_f16;
  long long dw_stride_tap, dw_stride_cw, dw_stride_k;
};

template <int TH_>   // tile height: 32 / 16 on large maps (fewer barriers and halo stagings per pixel), 8 otherwise
__global__ __launch_bounds__(256) void wgrad_direct_kernel(WGArgs a) {
  const int NC = a.CW / 8;
  const int lc = threadIdx.x % NC, lp = threadIdx.x / NC;
  const int ppb = 256 / NC;
  const int k = blockIdx.y;
  const long long npix = (long long)a.N * a.H * a.W;
  const int pad = (a.KS - 1) / 2;
  const int cpg = a.prologue ? a.CW / a.groups : 1;
  float acc[9][8];
  float bsum[8];
  float nsum = 0.f;
  float sc[8], sh[8];
  int cur_n = -1;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) bsum[j] = 0.f;
  // "wide" is the centre of the stencil:  dW[tap][cw] = sum_p wide[p] * narrow[p - sgn*off_tap].
  // A block walks TH_ x 16-pixel tiles (tile = blockIdx.x, + gridDim.x, ...), accumulating in registers across tiles.
  // Per tile the narrow halo (10x18 scalars of channel k, zero outside the image) is staged in LDS once, so the inner
  // loop is: one 16-byte wide load (+ prologue, applied ONCE), 9 LDS reads, 72 FMAs -- no bounds checks, no 64-bit
  // strided addressing.  (The first version fetched the 9 narrow neighbours from global memory per pixel piece with
  // per-lane 64-bit index arithmetic: 0.5 TB/s.)
  constexpr int TW_ = 16, HW_ = TW_ + 2, NPH = (TH_ + 2) * HW_;
  __shared__ float nar[2][NPH];
  const int tiles_x = (a.W + TW_ - 1) / TW_, tiles_y = (a.H + TH_ - 1) / TH_;
  const int ntiles = a.N * tiles_x * tiles_y;
  const int ITER = (TH_ * TW_ * NC + 255) / 256;          // pixel pieces of a tile per thread (NC <= 32 => <= 16)
  auto tile_org = [&](int tile, int& n, int& oy0, int& ox0) {
    const int tx = tile % tiles_x;
    const int r = tile / tiles_x;
    n = r / tiles_y;
    oy0 = (r - n * tiles_y) * TH_;
    ox0 = tx * TW_;
  };
  auto load_wide = [&](int tile, int it) -> u32x4 {
    int n, oy0, ox0;
    tile_org(tile, n, oy0, ox0);
    const int pi = it * ppb + lp;
    const int oy = oy0 + pi / TW_, ox = ox0 + pi % TW_;
    u32x4 r = u32x4{0u, 0u, 0u, 0u};
    if (tile < ntiles && pi < TH_ * TW_ && oy < a.H && ox < a.W)
      r = *(const u32x4*)(a.wide + ((size_t)(n * a.H + oy) * a.W + ox) * a.CW + lc * 8);
    return r;
  };
  constexpr int MAXIT = 4;    // ITER <= 4 for CW >= 64... handled in chunks of MAXIT below
  int buf = 0;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, buf ^= 1) {
    int n, oy0, ox0;
    tile_org(tile, n, oy0, ox0);
    // stage the narrow halo of channel k (threads 0..NPH-1), accumulate its interior sum for the narrow-side bias
    for (int hi = threadIdx.x; hi < NPH; hi += 256) {   // (18 x 18 = 324 halo values: two rounds)
      const int hy = hi / HW_, hx = hi - hy * HW_;
      const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
      float v = 0.f;
      if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) {
        v = ld_narrow(a.narrow, n * a.ns[0] + iy * a.ns[1] + ix * a.ns[2] + k * a.ns[3], a.narrow_f32);
        if (hy >= 1 && hy <= TH_ && hx >= 1 && hx <= TW_) nsum += v;
      }
      nar[buf][hi] = v;
    }
    if (a.prologue && n != cur_n) {
      cur_n = n;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ch = lc * 8 + j, g = ch / cpg;
        const float sum = stat_f(a.in_stats, (n * a.groups + g) * 2), sq = stat_f(a.in_stats, (n * a.groups + g) * 2 + 1);
        const float mean = sum * a.inv_cnt;
        const float rstd = rsqrtf(fmaxf(sq * a.inv_cnt - mean * mean, 0.f) + a.eps);
        sc[j] = rstd * a.gamma[ch];
        sh[j] = a.beta[ch] - mean * sc[j];
      }
    }
    __syncthreads();   // one barrier per tile: the other buffer is only rewritten after the next barrier
    for (int it0 = 0; it0 < ITER; it0 += MAXIT) {
      u32x4 raw[MAXIT];
#pragma unroll
      for (int u = 0; u < MAXIT; ++u) raw[u] = load_wide(tile, it0 + u);
#pragma unroll
      for (int u = 0; u < MAXIT; ++u) {
        const int pi = (it0 + u) * ppb + lp;
        const int py = pi / TW_, px = pi % TW_;
        if (it0 + u >= ITER || pi >= TH_ * TW_ || oy0 + py >= a.H || ox0 + px >= a.W) continue;
        float f[8];
        unpack8f(raw[u], f, a.wide_f16);
        if (a.dbias_wide && k == 0) {
#pragma unroll
          for (int j = 0; j < 8; ++j) bsum[j] += f[j];
        }
        if (a.prologue) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float v = f[j] * sc[j] + sh[j];
            if (a.prologue == PTI_PRO_GN_SILU) v = silu_f(v);
            f[j] = v;
          }
        }
        const float* nb = &nar[buf][(py + 1) * HW_ + px + 1];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            if (kh < a.KS && kw < a.KS) {
              const float nv = nb[-a.sgn * ((kh - pad) * HW_ + (kw - pad))];
#pragma unroll
              for (int j = 0; j < 8; ++j) acc[kh * 3 + kw][j] += nv * f[j];
            }
          }
        }
      }
    }
  }
  (void)npix;
  // reduction: (1) across the lanes of a wave that share lc (stride NC) by shuffles, (2) across the
  // 4 waves through LDS, (3) one atomicAdd per output element and block.
  extern __shared__ float red[];  // [4][80][NC]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ntap = a.KS * a.KS;
  auto wred = [&](float v) {
    for (int o = 32; o >= NC; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
  };
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = wred(acc[t][j]);
      if (lane < NC) red[(wave * 80 + t * 8 + j) * NC + lane] = v;
    }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = wred(bsum[j]);
    if (lane < NC) red[(wave * 80 + 72 + j) * NC + lane] = v;
  }
  __syncthreads();
  // per-block partials with plain stores; wgrad_direct_finalize sums them in block order (deterministic, and
  // no thousands of atomics on the same few hundred addresses)
  float* mine = a.part + ((size_t)k * gridDim.x + blockIdx.x) * (80 * NC + 1);
  for (int e = threadIdx.x; e < 80 * NC; e += 256)
    mine[e] = red[e] + red[80 * NC + e] + red[2 * 80 * NC + e] + red[3 * 80 * NC + e];
  {
    float v = nsum;   // every staging thread summed distinct interior pixels
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) mine[80 * NC] = red[0] + red[1] + red[2] + red[3];
  }
}

// out(+=) sum over blocks of the partials written by wgrad_direct_kernel: 64 consecutive elements per workgroup, the 16
// waves stride the block rows (8 loads in flight each) and fold through LDS in a fixed order (one thread per element
// walking all ~768 rows took 26 us of serial round trips)
__global__ __launch_bounds__(1024) void wgrad_direct_finalize_kernel(WGArgs a, int nblocks) {
  __shared__ float red[16][64];
  const int NC = a.CW / 8, k = blockIdx.y;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + lane;
  const int per = 80 * NC + 1;
  float acc = 0.f;
  if (e < per) {
    const float* base = a.part + (size_t)k * nblocks * per + e;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int b = w;
    for (; b + 7 * 16 < nblocks; b += 8 * 16) {
#pragma unroll
      for (int u = 0; u < 8; ++u) s[u] += base[(size_t)(b + 16 * u) * per];
    }
    for (; b < nblocks; b += 16) s[0] += base[(size_t)b * per];
    acc = ((s[0] + s[4]) + (s[1] + s[5])) + ((s[2] + s[6]) + (s[3] + s[7]));
  }
  red[w][lane] = acc;
  __syncthreads();
  if (w != 0 || e >= per) return;
  float v = red[0][lane];
#pragma unroll
  for (int g = 1; g < 16; ++g) v += red[g][lane];
  const int ntap = a.KS * a.KS;
  if (e == 80 * NC) {
    if (a.dbias_narrow) a.dbias_narrow[k] += v;
    return;
  }
  const int vi = e / NC, c = e % NC;
  if (vi < 72) {
    const int t = vi / 8, j = vi % 8;
    if (t < ntap) a.dw[t * a.dw_stride_tap + (c * 8 + j) * a.dw_stride_cw + k * a.dw_stride_k] += v;
  } else if (a.dbias_wide && k == 0) {
    a.dbias_wide[c * 8 + (vi - 72)] += v;
  }
}

}  // namespace

static int fill_common(DArgs& a, const void* x, const float* w, const float* bias, const int64_t* st, const float* g,
                       const float* b, void* y, const pti_conv_desc* d) {
  a.x = x; a.w = w; a.bias = bias; a.in_stats = (const stat_t*)st; a.gamma = g; a.beta = b; a.y = y;
  a.N = d->n; a.H = d->h; a.W = d->w; a.Cin = d->cin; a.Cout = d->cout; a.KS = d->ksize;
  a.prologue = d->prologue; a.groups = d->groups; a.eps = d->eps;
  a.inv_cnt = d->prologue ? 1.0f / ((float)(d->cin / d->groups) * (float)d->h * (float)d->w) : 0.f;
  a.in_f32 = d->in_f32; a.out_f32 = d->out_f32; a.w_lds = 0; a.wide_f16 = 0;
  for (int i = 0; i < 4; ++i) { a.is[i] = d->in_stride[i]; a.os[i] = d->out_stride[i]; }
  return 0;
}

// ---- derived operands of the degenerate-channel convs, all in ONE launch per optimiser step -----------------------------
// (as torch ops this was three to six tiny kernels per conv at the head of every step: ~17 launches on the critical path)
namespace {
__global__ __launch_bounds__(256) void direct_repack_kernel(pti_direct_repack_table t) {
  const pti_direct_repack_entry& e = t.e[blockIdx.y];
  const int total = e.cout * e.cin * 9;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int tap = i % 9, ci = (i / 9) % e.cin, co = i / (9 * e.cin);
    const float v = e.w[i];
    if (e.w_tck) e.w_tck[(tap * e.cin + ci) * e.cout + co] = v;             // [tap][ci][co]
    if (e.w_tck_t) e.w_tck_t[((8 - tap) * e.cout + co) * e.cin + ci] = v;     // data-gradient operand [8 - tap][co][ci]
    if (e.wpad) e.wpad[(co * e.pad_cin + ci) * 9 + tap] = v;                  // zero-padded master copy [cout'][pad_cin][9]
  }
  if (e.bpad && blockIdx.x == 0)
    for (int i = threadIdx.x; i < e.cout; i += 256) e.bpad[i] = e.b[i];
}
}  // namespace

extern "C" int pti_direct_repack(const pti_direct_repack_table* t, pti_stream_t s) {
  if (!t || t->n < 1 || t->n > PTI_DIRECT_REPACK_MAX) PTI_FAIL(PTI_EINVAL, "direct_repack: 1..%d entries", PTI_DIRECT_REPACK_MAX);
  int most = 0;
  for (int i = 0; i < t->n; ++i) {
    const pti_direct_repack_entry& e = t->e[i];
    if (!e.w || e.cout <= 0 || e.cin <= 0 || (e.wpad && e.pad_cin < e.cin) || (e.bpad && !e.b))
      PTI_FAIL(PTI_EINVAL, "direct_repack: entry %d", i);
    most = e.cout * e.cin * 9 > most ? e.cout * e.cin * 9 : most;
  }
  int blocks = (most + 255) / 256;
  if (blocks > 64) blocks = 64;
  PTI_LAUNCH(direct_repack_kernel, dim3(blocks, t->n), dim3(256), 0, (hipStream_t)s, *t);
  PTI_CHECK_LAUNCH("direct_repack");
  return PTI_OK;
}

extern "C" int pti_conv2d_direct(const void* x, const float* w, const float* bias, const int64_t* in_stats,
                                 const float* gamma, const float* beta, void* y, const pti_conv_desc* d,
                                 pti_stream_t s) {
  if (!x || !w || !y || !d) PTI_FAIL(PTI_EINVAL, "conv2d_direct: null pointer");
  if (d->mode != PTI_CONV_S1 || (d->ksize != 1 && d->ksize != 3)) PTI_FAIL(PTI_EUNSUPPORTED, "conv2d_direct: stride-1 k in {1,3} only");
  if (d->ho != d->h || d->wo != d->w || d->n <= 0 || d->h <= 0 || d->w <= 0) PTI_FAIL(PTI_EINVAL, "conv2d_direct: bad dims");
  DArgs a;
  fill_common(a, x, w, bias, in_stats, gamma, beta, y, d);
  const long long npix = (long long)d->n * d->h * d->w;
  if (d->cout % 32 == 0 && d->cin <= 16) {  // few cin -> many cout
    if (d->prologue) PTI_FAIL(PTI_EUNSUPPORTED, "conv2d_direct: prologue on the narrow input");
    if (d->out_f32) PTI_FAIL(PTI_EUNSUPPORTED, "conv2d_direct: wide output must be 16-bit NHWC");
    a.wide_f16 = d->out_f16;
    if (npix * 4 >= (1ll << 31)) PTI_FAIL(PTI_EUNSUPPORTED, "conv2d_direct: %lld pixels (the few-cin kernel indexes with 32 bits)", npix);
    long long gx = (npix * 4 + 255) / 256;   // four lanes (channel octets) per pixel; at most 4096 persistent workgroups
    static const long long cap = getenv("PTI_FEWCIN_WGS") ? atoll(getenv("PTI_FEWCIN_WGS")) : 4096;
    if (gx > cap) gx = cap;
    dim3 grid((unsigned)gx, d->cout / 32);
    PTI_LAUNCH(direct_fewcin_kernel, grid, dim3(256), 0, (hipStream_t)s, a);
  } else if (d->cout <= 16 && d->cin % 8 == 0 && d->cin >= 8 && d->cin <= 512 && !(d->cin & (d->cin - 1))) {
    if (d->in_f32) PTI_FAIL(PTI_EUNSUPPORTED, "conv2d_direct: wide input must be 16-bit NHWC");
    a.wide_f16 = d->in_f16;
    if (d->prologue && (!in_stats || !gamma || !beta || d->groups <= 0 || d->cin % d->groups))
      PTI_FAIL(PTI_EINVAL, "conv2d_direct: prologue needs stats/gamma/beta");
    if (d->cin < 32) PTI_FAIL(PTI_EUNSUPPORTED, "conv2d_direct: few-cout path needs cin >= 32");
    const int nc = d->cin / 8;
    auto launch = [&](auto cfg, auto kern) {
      using C = decltype(cfg);
      const int tiles_x = cdiv(d->w, C::TW), tiles_y = cdiv(d->h, C::TH);
      const size_t pbytes = (size_t)C::NP * C::NV * sizeof(float);
      const size_t wbytes = d->cout > 1 ? (size_t)d->ksize * d->ksize * nc * (8 * d->cout + 4) * sizeof(float) : 0;
      a.w_lds = (wbytes > 0 && pbytes + wbytes <= 64 * 1024) ? 1 : 0;   // else the weights are read through L1/L2
      const size_t lds = pbytes + (a.w_lds ? wbytes : 0);
      long long blocks = (long long)d->n * tiles_x * tiles_y;
      if (blocks > 1024) blocks = 1024;
      PTI_LAUNCH(kern, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)s, a, tiles_x, tiles_y);
    };
    if (d->cout == 1) launch(FoCfg<1>{}, direct_fewcout_kernel<1>);
    else if (d->cout <= 4) launch(FoCfg<4>{}, direct_fewcout_kernel<4>);
    else launch(FoCfg<16>{}, direct_fewcout_kernel<16>);
  } else {
    PTI_FAIL(PTI_EUNSUPPORTED, "conv2d_direct: cin=%d cout=%d is not a degenerate-channel shape", d->cin, d->cout);
  }
  PTI_CHECK_LAUNCH("conv2d_direct");
  return PTI_OK;
}

extern "C" int pti_wgrad_direct(const void* wide, const void* narrow, float* dw, float* dbias_wide,
                                float* dbias_narrow, const int64_t* in_stats, const float* gamma, const float* beta,
                                int n, int h, int w, int cw, int cn, int ksize, int sgn, int prologue, int groups,
                                float eps, int narrow_f32, int wide_f16, const int64_t* narrow_stride, int64_t dw_stride_tap,
                                int64_t dw_stride_cw, int64_t dw_stride_k, void* workspace, int64_t workspace_bytes,
                                pti_stream_t s) {
  if (!wide || !narrow || !dw || !narrow_stride || !workspace) PTI_FAIL(PTI_EINVAL, "wgrad_direct: null pointer");
  if (cw % 8 || cw < 8 || cw > 256 || (cw & (cw - 1)) || cn <= 0 || (ksize != 1 && ksize != 3))
    PTI_FAIL(PTI_EUNSUPPORTED, "wgrad_direct: cw=%d cn=%d k=%d", cw, cn, ksize);
  if (prologue && (!in_stats || !gamma || !beta || groups <= 0 || cw % groups)) PTI_FAIL(PTI_EINVAL, "wgrad_direct: prologue args");
  WGArgs a;
  a.wide = (const bf16*)wide; a.narrow = narrow; a.dw = dw; a.dbias_wide = dbias_wide; a.dbias_narrow = dbias_narrow;
  a.in_stats = (const stat_t*)in_stats; a.gamma = gamma; a.beta = beta;
  a.N = n; a.H = h; a.W = w; a.CW = cw; a.KS = ksize; a.sgn = sgn;
  a.prologue = prologue; a.groups = groups; a.eps = eps;
  a.inv_cnt = prologue ? 1.0f / ((float)(cw / groups) * (float)h * (float)w) : 0.f;
  a.narrow_f32 = narrow_f32; a.wide_f16 = wide_f16;
  for (int i = 0; i < 4; ++i) a.ns[i] = narrow_stride[i];
  a.dw_stride_tap = dw_stride_tap; a.dw_stride_cw = dw_stride_cw; a.dw_stride_k = dw_stride_k;
  const long long npix = (long long)n * h * w;
  if (npix * cw >= (1ll << 31)) PTI_FAIL(PTI_EUNSUPPORTED, "wgrad_direct: tensor too large for 32-bit pixel indexing");
  const int ppb = 256 / (cw / 8);
  // persistent blocks over 8x16-pixel tiles: 3 blocks/CU (160 VGPRs) x 256 CUs resident, >= 4 tiles each -- every
  // block writes 80*NC+1 partials, which for small maps would otherwise exceed the tensor itself
  // the tallest tile that still gives every block >= 4 tiles (256^2 x 32 channels, batch 32: 8 rows 108 us, 16 rows 91, 32 rows 82)
  int th = 8;
  for (int t = 32; t >= 16; t >>= 1)
    if ((long long)n * ((h + t - 1) / t) * ((w + 15) / 16) >= 4 * 768) { th = t; break; }
  const long long ntiles = (long long)n * ((h + th - 1) / th) * ((w + 15) / 16);
  long long blocks = ntiles / 4;
  if (blocks > 768) blocks = 768;
  if (blocks < 1) blocks = 1;
  const long long per = 80 * (cw / 8) + 1;
  while (blocks > 1 && blocks * cn * per * 4 > workspace_bytes) blocks /= 2;
  if (blocks * cn * per * 4 > workspace_bytes) PTI_FAIL(PTI_EINVAL, "wgrad_direct: workspace too small");
  a.part = (float*)workspace;
  if (th == 32)
    PTI_LAUNCH(wgrad_direct_kernel<32>, dim3((unsigned)blocks, cn), dim3(256), 4 * 80 * (cw / 8) * sizeof(float), (hipStream_t)s, a);
  else if (th == 16)
    PTI_LAUNCH(wgrad_direct_kernel<16>, dim3((unsigned)blocks, cn), dim3(256), 4 * 80 * (cw / 8) * sizeof(float), (hipStream_t)s, a);
  else
    PTI_LAUNCH(wgrad_direct_kernel<8>, dim3((unsigned)blocks, cn), dim3(256), 4 * 80 * (cw / 8) * sizeof(float), (hipStream_t)s, a);
  PTI_CHECK_LAUNCH("wgrad_direct");
  PTI_LAUNCH(wgrad_direct_finalize_kernel, dim3((unsigned)((per + 63) / 64), cn), dim3(1024), 0, (hipStream_t)s, a,
                     (int)blocks);
  PTI_CHECK_LAUNCH("wgrad_direct");
  return PTI_OK;
}
