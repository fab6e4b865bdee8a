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
  const float* in_stats;
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

// ---- few input channels -> many output channels (cout % 32 == 0, output NHWC bf16 dense) ----
// thread = (pixel, 32-cout block = blockIdx.y); weights are block-uniform (scalar loads).
__global__ __launch_bounds__(256) void direct_fewcin_kernel(DArgs a) {
  const long long pix = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long npix = (long long)a.N * a.H * a.W;
  if (pix >= npix) return;
  const int ox = pix % a.W;
  const int oy = (pix / a.W) % a.H;
  const int n = pix / ((long long)a.W * a.H);
  const int cb = blockIdx.y * 32;
  const int pad = (a.KS - 1) / 2;
  float acc[32];
#pragma unroll
  for (int c = 0; c < 32; ++c) acc[c] = a.bias ? a.bias[cb + c] : 0.f;
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
        const float* wr = wt + (size_t)ci * a.Cout;
#pragma unroll
        for (int c = 0; c < 32; ++c) acc[c] += v * wr[c];
      }
    }
  }
  bf16* yo = (bf16*)a.y + (size_t)pix * a.Cout + cb;
#pragma unroll
  for (int c = 0; c < 32; c += 8) *(u32x4*)(yo + c) = pack8f(acc + c, a.wide_f16);
}

// ---- many input channels (NHWC bf16 dense, cin % 8 == 0, cin/8 | 64) -> few output channels ----
// thread = (pixel, 8-channel piece); NC = cin/8 consecutive lanes cooperate on one pixel.
template <int MAXCO>
__global__ __launch_bounds__(256) void direct_fewcout_kernel(DArgs a) {
  const int NC = a.Cin / 8;
  const int lc = threadIdx.x % NC, lp = threadIdx.x / NC;
  const int ppb = 256 / NC;
  const long long npix = (long long)a.N * a.H * a.W;
  const int pad = (a.KS - 1) / 2;
  const int cpg = a.prologue ? a.Cin / a.groups : 1;
  const bf16* X = (const bf16*)a.x;
  // single-output-channel layers (decoder conv_out at full resolution) keep their 9x8 weights in registers
  extern __shared__ float wsm[];   // [k*k][Cin][Cout] when it fits (dynamic LDS size > 0)
  const int wcount = a.KS * a.KS * a.Cin * a.Cout;
  const bool w_in_lds = (MAXCO != 1) && (a.w_lds != 0);
  if (w_in_lds) {
    for (int i = threadIdx.x; i < wcount; i += 256) wsm[i] = a.w[i];
    __syncthreads();
  }
  const float* W = w_in_lds ? wsm : a.w;
  float wreg[MAXCO == 1 ? 9 : 1][8];
  if constexpr (MAXCO == 1) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) wreg[t][j] = (t < a.KS * a.KS) ? a.w[(size_t)(t * a.Cin + lc * 8 + j)] : 0.f;
  }
  float sc[8], sh[8];
  int cur_n = -1;
  for (long long pix0 = (long long)blockIdx.x * ppb; pix0 < npix; pix0 += (long long)gridDim.x * ppb) {
    const long long pix = pix0 + lp;
    const bool act = pix < npix;
    const long long pc = act ? pix : 0;
    const int ox = pc % a.W;
    const int oy = (pc / a.W) % a.H;
    const int n = pc / ((long long)a.W * a.H);
    if (a.prologue && n != cur_n) {
      cur_n = n;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ch = lc * 8 + j, g = ch / cpg;
        const float sum = a.in_stats[(n * a.groups + g) * 2], sq = a.in_stats[(n * a.groups + g) * 2 + 1];
        const float mean = sum * a.inv_cnt;
        const float rstd = rsqrtf(fmaxf(sq * a.inv_cnt - mean * mean, 0.f) + a.eps);
        sc[j] = rstd * a.gamma[ch];
        sh[j] = a.beta[ch] - mean * sc[j];
      }
    }
    float acc[MAXCO];
#pragma unroll
    for (int c = 0; c < MAXCO; ++c) acc[c] = 0.f;
    u32x4 raw[9];
    bool ok[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int kh = t / 3, kw = t % 3;
      const int iy = oy + kh - pad, ix = ox + kw - pad;
      ok[t] = act && kh < a.KS && kw < a.KS && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      raw[t] = u32x4{0u, 0u, 0u, 0u};
      if (ok[t]) raw[t] = *(const u32x4*)(X + ((size_t)(n * a.H + iy) * a.W + ix) * a.Cin + lc * 8);
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      if (!ok[t]) continue;
      float f[8];
      unpack8f(raw[t], f, a.wide_f16);
      if (a.prologue) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float v = f[j] * sc[j] + sh[j];
          if (a.prologue == PTI_PRO_GN_SILU) v = silu_f(v);
          f[j] = v;
        }
      }
      if constexpr (MAXCO == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[0] += f[j] * wreg[t][j];
      } else {
        const float* wt = W + ((size_t)(a.KS == 3 ? t : 0) * a.Cin + lc * 8) * a.Cout;
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
          for (int c = 0; c < MAXCO; ++c)
            if (c < a.Cout) acc[c] += f[j] * wt[j * a.Cout + c];
      }
    }
    // reduce over the NC lanes of this pixel (NC is a power of two <= 64, lanes are consecutive)
#pragma unroll
    for (int c = 0; c < MAXCO; ++c) {
      if (c < a.Cout) {
        float v = acc[c];
        for (int o = NC >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        acc[c] = v;
      }
    }
    if (act && lc == 0) {
      const long long ob = n * a.os[0] + oy * a.os[1] + ox * a.os[2];
#pragma unroll
      for (int c = 0; c < MAXCO; ++c)
        if (c < a.Cout) st_narrow(a.y, ob + c * a.os[3], a.out_f32, acc[c] + (a.bias ? a.bias[c] : 0.f));
    }
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
  const float* in_stats; const float* gamma; const float* beta;
  int N, H, W, CW, KS, sgn;
  int prologue, groups; float eps, inv_cnt;
  int narrow_f32; long long ns[4];
  int wide_f16;
  long long dw_stride_tap, dw_stride_cw, dw_stride_k;
};

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
  // "wide" is the centre of the stencil: one 16-byte load (+ prologue, applied ONCE) per pixel piece, multiplied with
  // the K*K neighbouring narrow scalars (4-byte, cache resident):  dW[tap][cw] = sum_p' wide[p'] * narrow[p' - sgn*off]
  for (long long pix0 = (long long)blockIdx.x * ppb; pix0 < npix; pix0 += (long long)gridDim.x * ppb) {
    const long long pix = pix0 + lp;
    if (pix >= npix) continue;
    const int ox = pix % a.W;
    const int oy = (pix / a.W) % a.H;
    const int n = pix / ((long long)a.W * a.H);
    if (a.prologue && n != cur_n) {
      cur_n = n;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ch = lc * 8 + j, g = ch / cpg;
        const float sum = a.in_stats[(n * a.groups + g) * 2], sq = a.in_stats[(n * a.groups + g) * 2 + 1];
        const float mean = sum * a.inv_cnt;
        const float rstd = rsqrtf(fmaxf(sq * a.inv_cnt - mean * mean, 0.f) + a.eps);
        sc[j] = rstd * a.gamma[ch];
        sh[j] = a.beta[ch] - mean * sc[j];
      }
    }
    float f[8];
    unpack8f(*(const u32x4*)(a.wide + (size_t)pix * a.CW + lc * 8), f, a.wide_f16);
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
    nsum += ld_narrow(a.narrow, n * a.ns[0] + oy * a.ns[1] + ox * a.ns[2] + k * a.ns[3], a.narrow_f32);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        if (kh < a.KS && kw < a.KS) {
          const int iy = oy - a.sgn * (kh - pad), ix = ox - a.sgn * (kw - pad);
          if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) {
            const float nv = ld_narrow(a.narrow, n * a.ns[0] + iy * a.ns[1] + ix * a.ns[2] + k * a.ns[3], a.narrow_f32);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[kh * 3 + kw][j] += nv * f[j];
          }
        }
      }
    }
  }
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
    float v = (lc == 0) ? nsum : 0.f;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) mine[80 * NC] = red[0] + red[1] + red[2] + red[3];
  }
}

// out(+=) sum over blocks of the partials written by wgrad_direct_kernel
__global__ __launch_bounds__(256) void wgrad_direct_finalize_kernel(WGArgs a, int nblocks) {
  const int NC = a.CW / 8, k = blockIdx.y;
  const int e = blockIdx.x * 256 + threadIdx.x;
  const int per = 80 * NC + 1;
  if (e >= per) return;
  const float* base = a.part + (size_t)k * nblocks * per + e;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f, s6 = 0.f, s7 = 0.f;
  int b = 0;
  for (; b + 7 < nblocks; b += 8) {
    s0 += base[(size_t)b * per];
    s1 += base[(size_t)(b + 1) * per];
    s2 += base[(size_t)(b + 2) * per];
    s3 += base[(size_t)(b + 3) * per];
    s4 += base[(size_t)(b + 4) * per];
    s5 += base[(size_t)(b + 5) * per];
    s6 += base[(size_t)(b + 6) * per];
    s7 += base[(size_t)(b + 7) * per];
  }
  for (; b < nblocks; ++b) s0 += base[(size_t)b * per];
  s0 += s4; s1 += s5; s2 += s6; s3 += s7;
  const float v = (s0 + s1) + (s2 + s3);
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

static int fill_common(DArgs& a, const void* x, const float* w, const float* bias, const float* st, const float* g,
                       const float* b, void* y, const pti_conv_desc* d) {
  a.x = x; a.w = w; a.bias = bias; a.in_stats = st; a.gamma = g; a.beta = b; a.y = y;
  a.N = d->n; a.H = d->h; a.W = d->w; a.Cin = d->cin; a.Cout = d->cout; a.KS = d->ksize;
  a.prologue = d->prologue; a.groups = d->groups; a.eps = d->eps;
  a.inv_cnt = d->prologue ? 1.0f / ((float)(d->cin / d->groups) * (float)d->h * (float)d->w) : 0.f;
  a.in_f32 = d->in_f32; a.out_f32 = d->out_f32; a.w_lds = 0; a.wide_f16 = 0;
  for (int i = 0; i < 4; ++i) { a.is[i] = d->in_stride[i]; a.os[i] = d->out_stride[i]; }
  return 0;
}

extern "C" int pti_conv2d_direct(const void* x, const float* w, const float* bias, const float* in_stats,
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
    dim3 grid((unsigned)((npix + 255) / 256), d->cout / 32);
    hipLaunchKernelGGL(direct_fewcin_kernel, grid, dim3(256), 0, (hipStream_t)s, a);
  } else if (d->cout <= 16 && d->cin % 8 == 0 && d->cin >= 8 && d->cin <= 512 && !(d->cin & (d->cin - 1))) {
    if (d->in_f32) PTI_FAIL(PTI_EUNSUPPORTED, "conv2d_direct: wide input must be 16-bit NHWC");
    a.wide_f16 = d->in_f16;
    if (d->prologue && (!in_stats || !gamma || !beta || d->groups <= 0 || d->cin % d->groups))
      PTI_FAIL(PTI_EINVAL, "conv2d_direct: prologue needs stats/gamma/beta");
    const int ppb = 256 / (d->cin / 8);
    long long blocks = (npix + ppb - 1) / ppb;
    if (blocks > 65536) blocks = 65536;
    if (blocks > 2048) blocks = 2048;
    const size_t wbytes = (size_t)d->ksize * d->ksize * d->cin * d->cout * sizeof(float);
    const size_t lds = (d->cout > 1 && wbytes <= 60 * 1024) ? wbytes : 0;
    a.w_lds = lds ? 1 : 0;
    if (lds && blocks > 512) blocks = 512;   // amortise the weight staging over several pixels per block
    if (d->cout == 1) hipLaunchKernelGGL(direct_fewcout_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, a);
    else if (d->cout <= 4) hipLaunchKernelGGL(direct_fewcout_kernel<4>, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)s, a);
    else hipLaunchKernelGGL(direct_fewcout_kernel<16>, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)s, a);
  } else {
    PTI_FAIL(PTI_EUNSUPPORTED, "conv2d_direct: cin=%d cout=%d is not a degenerate-channel shape", d->cin, d->cout);
  }
  PTI_CHECK_LAUNCH("conv2d_direct");
  return PTI_OK;
}

extern "C" int pti_wgrad_direct(const void* wide, const void* narrow, float* dw, float* dbias_wide,
                                float* dbias_narrow, const float* in_stats, const float* gamma, const float* beta,
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
  a.in_stats = in_stats; a.gamma = gamma; a.beta = beta;
  a.N = n; a.H = h; a.W = w; a.CW = cw; a.KS = ksize; a.sgn = sgn;
  a.prologue = prologue; a.groups = groups; a.eps = eps;
  a.inv_cnt = prologue ? 1.0f / ((float)(cw / groups) * (float)h * (float)w) : 0.f;
  a.narrow_f32 = narrow_f32; a.wide_f16 = wide_f16;
  for (int i = 0; i < 4; ++i) a.ns[i] = narrow_stride[i];
  a.dw_stride_tap = dw_stride_tap; a.dw_stride_cw = dw_stride_cw; a.dw_stride_k = dw_stride_k;
  const long long npix = (long long)n * h * w;
  const int ppb = 256 / (cw / 8);
  long long blocks = (npix + ppb - 1) / ppb;
  if (blocks > 1024) blocks = 1024;
  const long long per = 80 * (cw / 8) + 1;
  while (blocks > 1 && blocks * cn * per * 4 > workspace_bytes) blocks /= 2;
  if (blocks * cn * per * 4 > workspace_bytes) PTI_FAIL(PTI_EINVAL, "wgrad_direct: workspace too small");
  a.part = (float*)workspace;
  hipLaunchKernelGGL(wgrad_direct_kernel, dim3((unsigned)blocks, cn), dim3(256), 4 * 80 * (cw / 8) * sizeof(float),
                     (hipStream_t)s, a);
  PTI_CHECK_LAUNCH("wgrad_direct");
  hipLaunchKernelGGL(wgrad_direct_finalize_kernel, dim3((unsigned)((per + 255) / 256), cn), dim3(256), 0, (hipStream_t)s, a,
                     (int)blocks);
  PTI_CHECK_LAUNCH("wgrad_direct");
  return PTI_OK;
}
