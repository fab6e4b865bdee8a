// GroupNorm statistics and backward for NHWC bf16 tensors (gfx950).
//
// Replaces the reduction half of nn.GroupNorm (MONAI AEKLResBlock.norm1/norm2, the attention
// norm and the final Encoder/Decoder norm).  The apply half (+SiLU) is never
// a kernel of its own: it lives in the loader of the consuming convolution (conv_mfma.hip,
// conv_direct.hip).  Purely HBM-bound: every thread streams 16-byte pieces (8 channels).
#include "pti_common.h"

namespace {

// stats[n][g] += {sum, sumsq}.  grid = (blocks_per_sample, N), block = 256.
__global__ __launch_bounds__(256) void gn_stats_kernel(const bf16* __restrict__ x, stat_t* __restrict__ stats,
                                                       int HW, int C, int G, int pix_per_block, int x_f16) {
  extern __shared__ stat_t sm_q[];  // [G][2] fixed-point
  stat_t* sm = sm_q;
  const int n = blockIdx.y;
  const int tid = threadIdx.x;
  const int NC = C / 8;
  const int ppi = 256 / NC;  // pixels per iteration (NC divides 256 for C in {32..2048} powers of two)
  const int lc = tid % NC, lp = tid / NC;
  for (int i = tid; i < 2 * G; i += 256) sm[i] = 0;
  __syncthreads();
  const int p0 = blockIdx.x * pix_per_block;
  const int p1 = min(p0 + pix_per_block, HW);
  float s[8], q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = q[j] = 0.f;
  const bf16* base = x + (size_t)n * HW * C + lc * 8;
  int p = p0 + lp;
  for (; p + 3 * ppi < p1; p += 4 * ppi) {
    u32x4 r[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) r[u] = *(const u32x4*)(base + (size_t)(p + u * ppi) * C);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float f[8];
      unpack8f(r[u], f, x_f16);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        s[j] += f[j];
        q[j] += f[j] * f[j];
      }
    }
  }
  for (; p < p1; p += ppi) {
    const u32x4 r = *(const u32x4*)(base + (size_t)p * C);
    float f[8];
    unpack8f(r, f, x_f16);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      s[j] += f[j];
      q[j] += f[j] * f[j];
    }
  }
  const int cpg = C / G;
#pragma unroll
  for (int j = 0; j < 8; ++j)
    for (int o = 32; o >= NC; o >>= 1) {   // fold the lanes of this wave that hold the same channels
      s[j] += __shfl_xor(s[j], o, 64);
      q[j] += __shfl_xor(q[j], o, 64);
    }
  if ((tid & 63) < NC || NC > 64) {
    if (cpg >= 8) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) { a += s[j]; b += q[j]; }
      const int g = (lc * 8) / cpg;
      stat_add(&sm[2 * g], a);
      stat_add(&sm[2 * g + 1], b);
    } else {
      for (int j0 = 0; j0 < 8; j0 += cpg) {
        float a = 0.f, b = 0.f;
        for (int j = j0; j < j0 + cpg; ++j) { a += s[j]; b += q[j]; }
        const int g = (lc * 8 + j0) / cpg;
        stat_add(&sm[2 * g], a);
        stat_add(&sm[2 * g + 1], b);
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < 2 * G; i += 256)
    atomicAdd((unsigned long long*)&stats[(size_t)n * G * 2 + i], (unsigned long long)sm[i]);
}

}  // namespace

extern "C" int pti_gn_stats(const void* x, int64_t* stats, int n, int hw, int c, int groups, int x_f16, pti_stream_t s) {
  if (!x || !stats || n <= 0 || hw <= 0) PTI_FAIL(PTI_EINVAL, "gn_stats: bad pointer/dims");
  if (c < 8 || c > 2048 || (c & (c - 1)) || groups <= 0 || c % groups)
    PTI_FAIL(PTI_EUNSUPPORTED, "gn_stats: c=%d must be a power of two in [8,2048] and divisible by groups=%d", c, groups);
  const int cpg = c / groups;
  if (cpg < 8 && (8 % cpg)) PTI_FAIL(PTI_EUNSUPPORTED, "gn_stats: channels/group %d", cpg);
  if (cpg >= 8 && (cpg % 8)) PTI_FAIL(PTI_EUNSUPPORTED, "gn_stats: channels/group %d", cpg);
  const int ppi = 256 / (c / 8);
  // aim for ~2048 blocks over the whole launch, but never fewer than 16 iterations of ppi pixels per block
  // (small tensors: the per-block prologue/epilogue would dominate)
  int bps = cdiv(2048, n);
  int ppb = cdiv(hw, bps);
  if (ppb < 16 * ppi) ppb = 16 * ppi;
  ppb = cdiv(ppb, 4 * ppi) * 4 * ppi;
  bps = cdiv(hw, ppb);
  PTI_LAUNCH(gn_stats_kernel, dim3(bps, n), dim3(256), 2 * groups * sizeof(stat_t), (hipStream_t)s,
                     (const bf16*)x, (stat_t*)stats, hw, c, groups, ppb, x_f16);
  PTI_CHECK_LAUNCH("gn_stats");
  return PTI_OK;
}

// =============================================================================================
// Backward of  a = act(GroupNorm(x))  (act = SiLU or identity), NHWC bf16.
//   pass 1 (gn_bwd_reduce): per (sample, channel)  S1 = sum dy, S2 = sum dy*xhat  with
//           dy = dA * act'(y); also accumulates dbeta += S1, dgamma += S2 (atomics).
//   pass 2 (gn_bwd_apply):  dx = rstd*(gamma*dy - c1 - xhat*c2) [+ dres],  c1/c2 = group means of
//           gamma*dy and gamma*dy*xhat built from S1/S2.
// =============================================================================================
namespace {

struct GnbArgs {
  const bf16* x; const bf16* da; const bf16* dres; bf16* dx;
  const stat_t* stats; const float* gamma; const float* beta;
  float* sums;   // [N][C][2]
  float* dgamma; float* dbeta;
  int HW, C, G, silu, ppb;
  float eps, inv_cnt;
  int x_f16;   // x (a forward activation) is stored fp16; da / dres / dx are gradients: always bf16
  int N;
  float* part;   // gn_bwd_reduce: per-block partial sums [N][blocks per sample][C][2]
};

__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(GnbArgs a) {
  extern __shared__ float sm[];  // [4 waves][C][2]: every wave owns a slot, no atomics (bitwise reproducible)
  const int n = blockIdx.y, tid = threadIdx.x;
  const int NC = a.C / 8, ppi = 256 / NC, lc = tid % NC, lp = tid / NC;
  for (int i = tid; i < 8 * a.C; i += 256) sm[i] = 0.f;
  __syncthreads();
  const int cpg = a.C / a.G;
  float sc[8], sh[8], mu[8], rs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ch = lc * 8 + j, g = ch / cpg;
    const float sum = stat_f(a.stats, (n * a.G + g) * 2), sq = stat_f(a.stats, (n * a.G + g) * 2 + 1);
    const float mean = sum * a.inv_cnt;
    const float rstd = rsqrtf(fmaxf(sq * a.inv_cnt - mean * mean, 0.f) + a.eps);
    mu[j] = mean; rs[j] = rstd;
    sc[j] = rstd * a.gamma[ch];
    sh[j] = a.beta[ch] - mean * sc[j];
  }
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
  const int p0 = blockIdx.x * a.ppb, p1 = min(p0 + a.ppb, a.HW);
  const size_t base = (size_t)n * a.HW * a.C + lc * 8;
  constexpr int U = 4;  // independent 16-byte loads in flight per tensor and thread
  for (int pb = p0 + lp; pb < p1; pb += U * ppi) {
    u32x4 rx[U], rd[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = pb + u * ppi;
      rx[u] = rd[u] = u32x4{0u, 0u, 0u, 0u};
      if (p < p1) {
        rx[u] = *(const u32x4*)(a.x + base + (size_t)p * a.C);
        rd[u] = *(const u32x4*)(a.da + base + (size_t)p * a.C);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float fx[8], fd[8];
      unpack8f(rx[u], fx, a.x_f16);
      unpack8(rd[u], fd);   // da == 0 for the padded lanes => they add nothing
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float dy = fd[j];
        if (a.silu) dy *= dsilu_f(fx[j] * sc[j] + sh[j]);
        s1[j] += dy;
        s2[j] += dy * (fx[j] - mu[j]) * rs[j];
      }
    }
  }
  // lanes lc, lc+NC, lc+2NC, ... of a wave hold the same channels: fold them with shuffles first so
  // that only NC lanes per wave touch the LDS accumulators (256 contended LDS atomics -> 4 per address)
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float v1 = s1[j], v2 = s2[j];
    for (int o = 32; o >= NC; o >>= 1) {
      v1 += __shfl_xor(v1, o, 64);
      v2 += __shfl_xor(v2, o, 64);
    }
    // NC <= 64: after the fold lanes 0..NC-1 of each wave hold the wave's totals for their channels
    if ((tid & 63) < NC) {
      float* slot = sm + (tid >> 6) * 2 * a.C;
      slot[(lc * 8 + j) * 2] = v1;
      slot[(lc * 8 + j) * 2 + 1] = v2;
    }
  }
  __syncthreads();
  // block partial -> part[n][block][c][s] (plain stores; pti_gn_sums_finalize adds the blocks up in a fixed order;
  // dgamma / dbeta are folded from the finished sums by one block of the apply kernel)
  for (int i = tid; i < 2 * a.C; i += 256) {
    const float v = (sm[i] + sm[2 * a.C + i]) + (sm[4 * a.C + i] + sm[6 * a.C + i]);
    a.part[((size_t)n * gridDim.x + blockIdx.x) * 2 * a.C + i] = v;
  }
}

// sums[n][i] = sum_t part[n][t][i] for i = (channel, {sum dy, sum dy*xhat}): the data-gradient convs and
// gn_bwd_reduce write one partial row per pixel tile / block with plain stores, and this kernel adds the rows up in
// a fixed order (wave w takes tiles w, w+16, ...; the 16 wave totals are folded as a fixed tree), so the totals do
// not depend on the order workgroups ran in.  64 consecutive i per workgroup: every load is one 256-byte row piece.
__global__ __launch_bounds__(1024) void gn_sums_finalize_kernel(const float* __restrict__ part, float* __restrict__ sums,
                                                                int T, int R, const float* __restrict__ wsums,
                                                                float* __restrict__ wdgamma, float* __restrict__ wdbeta,
                                                                int wn, int wc) {
  __shared__ float red[16][64];
  const int n = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  // rider (pti_gn_sums_finalize_affine): the affine gradients of ANOTHER, already finalized GroupNorm whose backward was
  // applied inside a conv's loader -- dgamma[c] += sum_n wsums[n][c][1], dbeta[c] += sum_n wsums[n][c][0], samples in order
  if (wsums && blockIdx.x == 0 && n == 0)
    for (int i = threadIdx.x; i < wc; i += 1024) {
      float d1 = 0.f, d2 = 0.f;
      for (int m = 0; m < wn; ++m) {
        const f32x2 sv = *(const f32x2*)(wsums + ((size_t)m * wc + i) * 2);
        d1 += sv[0];
        d2 += sv[1];
      }
      if (wdbeta) wdbeta[i] += d1;
      if (wdgamma) wdgamma[i] += d2;
    }
  const int i = blockIdx.x * 64 + lane;
  float v = 0.f;
  if (i < R) {
    const float* p = part + (size_t)n * T * R + i;
    int t = w;
    for (; t + 48 < T; t += 64) {   // four independent loads in flight
      const float a0 = p[(size_t)t * R], a1 = p[(size_t)(t + 16) * R], a2 = p[(size_t)(t + 32) * R], a3 = p[(size_t)(t + 48) * R];
      v += (a0 + a1) + (a2 + a3);
    }
    for (; t < T; t += 16) v += p[(size_t)t * R];
  }
  red[w][lane] = v;
  __syncthreads();
  if (w == 0 && i < R) {
    float r[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) r[k] = red[k][lane];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1)
#pragma unroll
      for (int k = 0; k < o; ++k) r[k] += r[k + o];
    sums[(size_t)n * R + i] = r[0];
  }
}

__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(GnbArgs a) {
  const int n = blockIdx.y, tid = threadIdx.x;
  const int NC = a.C / 8, ppi = 256 / NC, lc = tid % NC, lp = tid / NC;
  const int cpg = a.C / a.G;
  // the first batch of tensor pieces is requested BEFORE the per-channel parameters (three dependent round trips of
  // statistics / gamma / sums): on the small maps a workgroup has only one or two batches and the parameter chain was
  // a third of the launch
  const int p0 = blockIdx.x * a.ppb, p1 = min(p0 + a.ppb, a.HW);
  const size_t base = (size_t)n * a.HW * a.C + lc * 8;
  constexpr int U = 4;
  u32x4 rx[U], rd[U], rr[U];
  auto load_batch = [&](int pb) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = pb + u * ppi;
      rx[u] = rd[u] = rr[u] = u32x4{0u, 0u, 0u, 0u};
      if (p < p1) {
        // streamed once: non-temporal loads and store (-0.6 % per step against the default cache policy)
        rx[u] = __builtin_nontemporal_load((const u32x4*)(a.x + base + (size_t)p * a.C));
        rd[u] = __builtin_nontemporal_load((const u32x4*)(a.da + base + (size_t)p * a.C));
        if (a.dres) rr[u] = __builtin_nontemporal_load((const u32x4*)(a.dres + base + (size_t)p * a.C));
      }
    }
  };
  if (p0 + lp < p1) load_batch(p0 + lp);
  float sc[8], sh[8], mu[8], rs[8], ga[8], c1[8], c2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ch = lc * 8 + j, g = ch / cpg;
    const float sum = stat_f(a.stats, (n * a.G + g) * 2), sq = stat_f(a.stats, (n * a.G + g) * 2 + 1);
    const float mean = sum * a.inv_cnt;
    const float rstd = rsqrtf(fmaxf(sq * a.inv_cnt - mean * mean, 0.f) + a.eps);
    mu[j] = mean; rs[j] = rstd; ga[j] = a.gamma[ch];
    sc[j] = rstd * ga[j];
    sh[j] = a.beta[ch] - mean * sc[j];
    c1[j] = c2[j] = 0.f;
    if (cpg > 8) {   // group spans several 8-channel pieces: walk it (rare: >8 channels per group)
      float t1 = 0.f, t2 = 0.f;
      for (int cc = g * cpg; cc < (g + 1) * cpg; ++cc) {
        const float gm = a.gamma[cc];
        t1 += gm * a.sums[((size_t)n * a.C + cc) * 2];
        t2 += gm * a.sums[((size_t)n * a.C + cc) * 2 + 1];
      }
      c1[j] = t1 * a.inv_cnt;
      c2[j] = t2 * a.inv_cnt;
    }
  }
  if (cpg <= 8) {    // whole groups live inside this thread's 8 channels: 16 loads, no dependent walk
    float w1[8], w2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const f32x2 sv = *(const f32x2*)(a.sums + ((size_t)n * a.C + lc * 8 + j) * 2);
      w1[j] = ga[j] * sv[0];
      w2[j] = ga[j] * sv[1];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const bool same = (jj / cpg) == (j / cpg);
        t1 += same ? w1[jj] : 0.f;
        t2 += same ? w2[jj] : 0.f;
      }
      c1[j] = t1 * a.inv_cnt;
      c2[j] = t2 * a.inv_cnt;
    }
  }
  if (blockIdx.x == 0 && n == 0) {  // affine gradients: ONE block walks the samples in order (no atomics: reproducible)
    for (int i = tid; i < a.C; i += 256) {
      float d1 = 0.f, d2 = 0.f;
      for (int m = 0; m < a.N; ++m) {
        d1 += a.sums[((size_t)m * a.C + i) * 2];
        d2 += a.sums[((size_t)m * a.C + i) * 2 + 1];
      }
      if (a.dbeta) a.dbeta[i] += d1;
      if (a.dgamma) a.dgamma[i] += d2;
    }
  }
  for (int pb = p0 + lp; pb < p1; pb += U * ppi) {
    if (pb != p0 + lp) load_batch(pb);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = pb + u * ppi;
      float fx[8], fd[8], fr[8], o[8];
      unpack8f(rx[u], fx, a.x_f16);
      unpack8(rd[u], fd);
      unpack8(rr[u], fr);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float dy = fd[j];
        if (a.silu) dy *= dsilu_f(fx[j] * sc[j] + sh[j]);
        const float xh = (fx[j] - mu[j]) * rs[j];
        o[j] = rs[j] * (ga[j] * dy - c1[j] - xh * c2[j]) + fr[j];
      }
      if (p < p1) __builtin_nontemporal_store(pack8(o), (u32x4*)(a.dx + base + (size_t)p * a.C));
    }
  }
}

// 2x2 sum pooling (backward of nearest 2x up-sampling): y[n,h,w,c] = sum of x[n,2h+{0,1},2w+{0,1},c]
__global__ __launch_bounds__(256) void pool2x2_sum_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, int N,
                                                          int H, int W, int C) {
  const int NC = C / 8;
  const long long total = (long long)N * H * W * NC;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int lc = e % NC;
    long long p = e / NC;
    const int w = p % W; p /= W;
    const int h = p % H;
    const int n = p / H;
    const bf16* src = x + (((size_t)n * 2 * H + 2 * h) * 2 * W + 2 * w) * C + lc * 8;
    float f0[8], f1[8], f2[8], f3[8], o[8];
    unpack8(*(const u32x4*)(src), f0);
    unpack8(*(const u32x4*)(src + C), f1);
    unpack8(*(const u32x4*)(src + (size_t)2 * W * C), f2);
    unpack8(*(const u32x4*)(src + (size_t)2 * W * C + C), f3);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (f0[j] + f1[j]) + (f2[j] + f3[j]);
    *(u32x4*)(y + (((size_t)n * H + h) * W + w) * C + lc * 8) = pack8(o);
  }
}

}  // namespace

// pixels per thread a block handles at least (amortises the per-thread GroupNorm parameter set-up).  8 measured
// best: 128 channels @32^2 (the smallest maps of config A) 18.4 -> 15.5 us against 16, the large maps unchanged; 4
// and 2 lose 15 % on 128 channels @64^2.  PTI_GNB_MIN_ITERS overrides (tuning knob).
static int gnb_min_iters() {
  static const int v = getenv("PTI_GNB_MIN_ITERS") ? atoi(getenv("PTI_GNB_MIN_ITERS")) : 8;
  return v < 1 ? 1 : v;
}
static int gn_bwd_grid(int n, int hw, int c, int* ppb_out) {
  const int ppi = 256 / (c / 8);
  int bps = cdiv(2048, n);
  int ppb = cdiv(hw, bps);
  if (ppb < gnb_min_iters() * ppi) ppb = gnb_min_iters() * ppi;
  ppb = cdiv(ppb, ppi) * ppi;
  if (ppb_out) *ppb_out = ppb;
  return cdiv(hw, ppb);
}

// blocks per sample of the reduction launch: `partials` of pti_gn_bwd holds n * c * 2 * blocks floats
extern "C" int pti_gn_bwd_blocks(int n, int hw, int c) {
  if (n <= 0 || hw <= 0 || c < 8 || (c & (c - 1)) || c > 512) return 0;
  return gn_bwd_grid(n, hw, c, nullptr);
}

extern "C" int pti_gn_bwd(const void* x, const void* da, const void* dres, void* dx, const int64_t* stats,
                          const float* gamma, const float* beta, float* sums, float* partials, float* dgamma,
                          float* dbeta, int n, int hw, int c, int groups, float eps, int silu, int x_f16, pti_stream_t s) {
  if (!x || !da || !dx || !stats || !gamma || !beta || !sums || !partials) PTI_FAIL(PTI_EINVAL, "gn_bwd: null pointer");
  if (c < 8 || c > 512 || (c & (c - 1)) || groups <= 0 || c % groups) PTI_FAIL(PTI_EUNSUPPORTED, "gn_bwd: c=%d groups=%d", c, groups);
  GnbArgs a;
  a.x = (const bf16*)x; a.da = (const bf16*)da; a.dres = (const bf16*)dres; a.dx = (bf16*)dx;
  a.stats = (const stat_t*)stats; a.gamma = gamma; a.beta = beta; a.sums = sums; a.dgamma = dgamma; a.dbeta = dbeta;
  a.HW = hw; a.C = c; a.G = groups; a.silu = silu; a.eps = eps; a.x_f16 = x_f16; a.N = n;
  a.inv_cnt = 1.0f / ((float)(c / groups) * (float)hw);
  int ppb;
  const int bps = gn_bwd_grid(n, hw, c, &ppb);
  a.ppb = ppb;
  a.part = partials;
  PTI_LAUNCH(gn_bwd_reduce_kernel, dim3(bps, n), dim3(256), 8 * c * sizeof(float), (hipStream_t)s, a);
  PTI_CHECK_LAUNCH("gn_bwd_reduce");
  PTI_LAUNCH(gn_sums_finalize_kernel, dim3(cdiv(2 * c, 64), n), dim3(1024), 0, (hipStream_t)s, partials, sums,
                     bps, 2 * c, (const float*)nullptr, (float*)nullptr, (float*)nullptr, 0, 0);
  PTI_CHECK_LAUNCH("gn_sums_finalize");
  PTI_LAUNCH(gn_bwd_apply_kernel, dim3(bps, n), dim3(256), 0, (hipStream_t)s, a);
  PTI_CHECK_LAUNCH("gn_bwd_apply");
  return PTI_OK;
}

// second half only: `dy` already is dA * act'(GN(x)) and `sums` already holds {sum dy, sum dy*xhat} per (n,c)
// (both produced by pti_conv2d_mfma_gnbwd in the data-gradient conv's epilogue)
extern "C" int pti_gn_bwd_apply(const void* x, const void* dy, const void* dres, void* dx, const int64_t* stats,
                                const float* gamma, const float* beta, const float* sums, float* dgamma, float* dbeta,
                                int n, int hw, int c, int groups, float eps, int x_f16, pti_stream_t s) {
  if (!x || !dy || !dx || !stats || !gamma || !beta || !sums) PTI_FAIL(PTI_EINVAL, "gn_bwd_apply: null pointer");
  if (c < 8 || c > 2048 || (c & (c - 1)) || groups <= 0 || c % groups) PTI_FAIL(PTI_EUNSUPPORTED, "gn_bwd_apply: c=%d groups=%d", c, groups);
  GnbArgs a;
  a.x = (const bf16*)x; a.da = (const bf16*)dy; a.dres = (const bf16*)dres; a.dx = (bf16*)dx;
  a.stats = (const stat_t*)stats; a.gamma = gamma; a.beta = beta; a.sums = const_cast<float*>(sums); a.dgamma = dgamma; a.dbeta = dbeta;
  a.HW = hw; a.C = c; a.G = groups; a.silu = 0; a.eps = eps; a.x_f16 = x_f16; a.N = n;
  a.inv_cnt = 1.0f / ((float)(c / groups) * (float)hw);
  const int ppi = 256 / (c / 8);
  int bps = cdiv(2048, n);
  int ppb = cdiv(hw, bps);
  if (ppb < gnb_min_iters() * ppi) ppb = gnb_min_iters() * ppi;
  ppb = cdiv(ppb, ppi) * ppi;
  bps = cdiv(hw, ppb);
  a.ppb = ppb;
  PTI_LAUNCH(gn_bwd_apply_kernel, dim3(bps, n), dim3(256), 0, (hipStream_t)s, a);
  PTI_CHECK_LAUNCH("gn_bwd_apply");
  return PTI_OK;
}

// dgamma[c] += sum_n sums[n][c][1], dbeta[c] += sum_n sums[n][c][0] (samples in order: reproducible) -- the affine
// gradients of a GroupNorm whose backward was applied inside the next data-gradient conv (pti_conv2d_mfma_gnbwd_chain),
// i.e. without the pti_gn_bwd_apply launch that otherwise produces them.
namespace {
__global__ __launch_bounds__(256) void gn_affine_grads_kernel(const float* __restrict__ sums, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int n, int c) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= c) return;
  float d1 = 0.f, d2 = 0.f;
  for (int m = 0; m < n; ++m) {
    const f32x2 sv = *(const f32x2*)(sums + ((size_t)m * c + i) * 2);
    d1 += sv[0];
    d2 += sv[1];
  }
  if (dbeta) dbeta[i] += d1;
  if (dgamma) dgamma[i] += d2;
}
}  // namespace

extern "C" int pti_gn_affine_grads(const float* sums, float* dgamma, float* dbeta, int n, int c, pti_stream_t s) {
  if (!sums || n <= 0 || c <= 0) PTI_FAIL(PTI_EINVAL, "gn_affine_grads: bad args");
  PTI_LAUNCH(gn_affine_grads_kernel, dim3(cdiv(c, 256)), dim3(256), 0, (hipStream_t)s, sums, dgamma, dbeta, n, c);
  PTI_CHECK_LAUNCH("gn_affine_grads");
  return PTI_OK;
}

extern "C" int pti_gn_sums_finalize(const float* partials, float* sums, int n, int tiles, int row_len, pti_stream_t s) {
  if (!partials || !sums || n <= 0 || tiles <= 0 || row_len <= 0) PTI_FAIL(PTI_EINVAL, "gn_sums_finalize: bad args");
  PTI_LAUNCH(gn_sums_finalize_kernel, dim3(cdiv(row_len, 64), n), dim3(1024), 0, (hipStream_t)s, partials, sums,
                     tiles, row_len, (const float*)nullptr, (float*)nullptr, (float*)nullptr, 0, 0);
  PTI_CHECK_LAUNCH("gn_sums_finalize");
  return PTI_OK;
}

// pti_gn_sums_finalize + the affine gradients of another GroupNorm (see pti_gn_affine_grads) in the same launch
extern "C" int pti_gn_sums_finalize_affine(const float* partials, float* sums, int n, int tiles, int row_len,
                                           const float* affine_sums, float* dgamma, float* dbeta, int affine_c, pti_stream_t s) {
  if (!partials || !sums || n <= 0 || tiles <= 0 || row_len <= 0 || !affine_sums || affine_c <= 0)
    PTI_FAIL(PTI_EINVAL, "gn_sums_finalize_affine: bad args");
  PTI_LAUNCH(gn_sums_finalize_kernel, dim3(cdiv(row_len, 64), n), dim3(1024), 0, (hipStream_t)s, partials, sums,
                     tiles, row_len, affine_sums, dgamma, dbeta, n, affine_c);
  PTI_CHECK_LAUNCH("gn_sums_finalize_affine");
  return PTI_OK;
}

extern "C" int pti_pool2x2_sum(const void* x, void* y, int n, int h, int w, int c, pti_stream_t s) {
  if (!x || !y || n <= 0 || h <= 0 || w <= 0 || c % 8) PTI_FAIL(PTI_EINVAL, "pool2x2_sum: bad args");
  const long long total = (long long)n * h * w * (c / 8);
  long long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  PTI_LAUNCH(pool2x2_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, (const bf16*)x, (bf16*)y, n, h, w, c);
  PTI_CHECK_LAUNCH("pool2x2_sum");
  return PTI_OK;
}
