// Small fused kernels around the latent bottleneck and the optimiser step (gfx950).
//
//  * pti_latent_head_fwd/bwd: MONAI AutoencoderKL.encode tail + sampling + post_quant_conv
//      mu = quant_conv_mu(h); lv = clamp(quant_conv_log_sigma(h), -30, 20); sigma = exp(lv/2)
//      z = mu + eps*sigma; zq = post_quant_conv(z)          
//    three 1x1 convs on <=16 channels + clamp + exp + FMA collapse into one elementwise kernel.
//  * pti_vae_loss: reconstruction (L1|L2 mean) + KL exactly as reference
//      src/pti_ldm_vae/models/losses.py:25-30 / vae_scripts/train_vae.py:393-394, producing the loss
//      scalars and the gradient seeds d(recon), d(mu), d(third) of  recon + kl_weight*kl  in one pass.
//  * pti_adam_step: torch.optim.Adam (train_vae.py:301, defaults) on the flat fp32 parameter arena.
#include "pti_common.h"

namespace {

constexpr int MAXL = 16;

struct LatArgs {
  const float* h;        // [B,HW,L] fp32 (NHWC)
  const float* eps;      // [B,L,HW] fp32 (NCHW) or null (=> z = mu)
  const float* wm; const float* bm; const float* wl; const float* bl; const float* wp; const float* bp;  // [L][L], [L]
  float* mu; float* sigma; float* logvar;   // [B,L,HW] fp32 (NCHW); logvar may be null
  float* zq;             // [B,HW,L] fp32 (NHWC)
  int B, HW, L;
};

__global__ __launch_bounds__(256) void latent_fwd_kernel(LatArgs a) {
  const long long total = (long long)a.B * a.HW;
  const int L = a.L;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int p = e % a.HW;
    const int b = e / a.HW;
    float hv[MAXL], z[MAXL];
#pragma unroll
    for (int j = 0; j < MAXL; ++j) hv[j] = (j < L) ? a.h[e * L + j] : 0.f;
#pragma unroll
    for (int i = 0; i < MAXL; ++i) {
      if (i < L) {
        float m = a.bm[i], l = a.bl[i];
#pragma unroll
        for (int j = 0; j < MAXL; ++j)
          if (j < L) {
            m += a.wm[i * L + j] * hv[j];
            l += a.wl[i * L + j] * hv[j];
          }
        l = fminf(fmaxf(l, -30.f), 20.f);
        const float sg = expf(0.5f * l);
        const size_t o = ((size_t)b * L + i) * a.HW + p;
        a.mu[o] = m;
        a.sigma[o] = sg;
        if (a.logvar) a.logvar[o] = l;
        z[i] = a.eps ? m + a.eps[o] * sg : m;
      }
    }
#pragma unroll
    for (int i = 0; i < MAXL; ++i) {
      if (i < L) {
        float v = a.bp[i];
#pragma unroll
        for (int j = 0; j < MAXL; ++j)
          if (j < L) v += a.wp[i * L + j] * z[j];
        a.zq[e * L + i] = v;
      }
    }
  }
}

// decode-only entry: zq = post_quant_conv(z) for a user-supplied z (NCHW fp32)
__global__ __launch_bounds__(256) void post_quant_kernel(const float* z, const float* wp, const float* bp, float* zq,
                                                         int B, int HW, int L) {
  const long long total = (long long)B * HW;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int p = e % HW;
    const int b = e / HW;
    for (int i = 0; i < L; ++i) {
      float v = bp[i];
      for (int j = 0; j < L; ++j) v += wp[i * L + j] * z[((size_t)b * L + j) * HW + p];
      zq[e * L + i] = v;
    }
  }
}

// backward of post_quant_kernel: dz[b][j][p] = sum_i wp[i][j] dzq[b][p][i]; block partials of gwp[i][j] = sum dzq_i z_j
// and gbp = sum dzq go to part[block][L*L+L] (added up in a fixed order by block_partials_finalize_kernel).
// LT > 0: L == LT at compile time, the partials live in registers and fold with shuffles (no atomics at all).
template <int LT>
__global__ __launch_bounds__(256) void post_quant_bwd_kernel(const float* dzq, const float* z, const float* wp, float* dz,
                                                             float* part, int B, int HW, int Lrt) {
  extern __shared__ float sm[];  // LT > 0: [4 waves][L*L + L], else L*L + L
  const int L = LT > 0 ? LT : Lrt, LL = L * L + L;
  constexpr int NACC = LT > 0 ? LT * LT + LT : 1;
  float racc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) racc[i] = 0.f;
  for (int i = threadIdx.x; i < LL; i += 256) sm[i] = 0.f;
  __syncthreads();
  const long long total = (long long)B * HW;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int p = e % HW;
    const int b = e / HW;
    float g[MAXL], zz[MAXL];
    for (int i = 0; i < L; ++i) {
      g[i] = dzq[e * L + i];
      zz[i] = z[((size_t)b * L + i) * HW + p];
    }
    for (int j = 0; j < L; ++j) {
      float v = 0.f;
      for (int i = 0; i < L; ++i) v += wp[i * L + j] * g[i];
      if (dz) dz[((size_t)b * L + j) * HW + p] = v;
    }
    for (int i = 0; i < L; ++i) {
      if constexpr (LT > 0) {
        racc[L * L + i] += g[i];
        for (int j = 0; j < L; ++j) racc[i * L + j] += g[i] * zz[j];
      } else {
        atomicAdd(&sm[L * L + i], g[i]);
        for (int j = 0; j < L; ++j) atomicAdd(&sm[i * L + j], g[i] * zz[j]);
      }
    }
  }
  if constexpr (LT > 0) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      const float v = wave_sum(racc[i]);
      if ((threadIdx.x & 63) == 0) sm[(threadIdx.x >> 6) * NACC + i] = v;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < LL; i += 256) {
    float v = sm[i];
    if constexpr (LT > 0) v = (sm[i] + sm[NACC + i]) + (sm[2 * NACC + i] + sm[3 * NACC + i]);
    part[(size_t)blockIdx.x * LL + i] = v;
  }
}

// Generic latent width (L <= 16, e.g. the AR config's 16): chunks of 128 elements; phase 1 = the per-element arithmetic,
// one element per thread (threads 0..127), operands of the parameter gradients parked in LDS; phase 2 = thread (i, j) of
// the L x L grid owns gwp[i][j] and sums its 128 products IN ELEMENT ORDER (fixed order: bitwise reproducible; the first
// version folded per-element partials with LDS float atomics -- 816 contended atomics per element at L = 16: 455 us
// for a 2 MB map, and run-to-run rounding differences).
constexpr int GCH = 128;
__global__ __launch_bounds__(256) void post_quant_bwd_generic_kernel(const float* dzq, const float* z, const float* wp, float* dz,
                                                                     float* part, int B, int HW, int L) {
  extern __shared__ float sm[];   // s_g [GCH][L], s_z [GCH][L]
  float* s_g = sm;
  float* s_z = sm + GCH * L;
  const int t = threadIdx.x, LL = L * L + L;
  const int pi = t / L, pj = t - pi * L;
  const bool pair = t < L * L;
  float aw = 0.f, ab = 0.f;
  const long long total = (long long)B * HW, nch = (total + GCH - 1) / GCH;
  for (long long c = blockIdx.x; c < nch; c += gridDim.x) {
    const long long e = c * GCH + t;
    if (t < GCH) {
      if (e < total) {
        const int p = e % HW;
        const int b = e / HW;
        float g[MAXL];
        for (int i = 0; i < L; ++i) {
          g[i] = dzq[e * L + i];
          s_g[t * L + i] = g[i];
          s_z[t * L + i] = z[((size_t)b * L + i) * HW + p];
        }
        if (dz)
          for (int j = 0; j < L; ++j) {
            float v = 0.f;
            for (int i = 0; i < L; ++i) v += wp[i * L + j] * g[i];
            dz[((size_t)b * L + j) * HW + p] = v;
          }
      } else {
        for (int i = 0; i < L; ++i) s_g[t * L + i] = s_z[t * L + i] = 0.f;
      }
    }
    __syncthreads();
    if (pair)
      for (int k = 0; k < GCH; ++k) aw += s_g[k * L + pi] * s_z[k * L + pj];
    if (t < L)
      for (int k = 0; k < GCH; ++k) ab += s_g[k * L + t];
    __syncthreads();
  }
  float* row = part + (size_t)blockIdx.x * LL;
  if (pair) row[pi * L + pj] = aw;
  if (t < L) row[L * L + t] = ab;
}

struct LatBwdArgs {
  const float* h; const float* eps;
  const float* wm; const float* bm; const float* wl; const float* bl; const float* wp; const float* bp;
  const float* dzq;      // [B,HW,L] NHWC fp32 or null
  const float* dmu;      // [B,L,HW] NCHW fp32 or null   (external gradient on mu)
  const float* dsigma;   // [B,L,HW] NCHW fp32 or null   (external gradient on sigma)
  float* dh;             // [B,HW,L]
  float* gwm; float* gbm; float* gwl; float* gbl; float* gwp; float* gbp;  // accumulated by the finalising launch
  int B, HW, L;
  float* part;           // [blocks][3*(L*L+L)] block partials of the six parameter gradients
};

// Fixed-order second stage of the small reductions (latent-head parameter gradients, loss terms): workgroup i adds
// value i of every block partial row -- lanes stride the rows, then a fixed shuffle tree -- and ONE lane accumulates
// it into its destination.  No float atomics: the result does not depend on the order the first stage's blocks ran.
struct FinSegs { float* dst[6]; int len[6]; int nseg; };
__global__ __launch_bounds__(64) void block_partials_finalize_kernel(const float* __restrict__ part, int nb, int stride,
                                                                     FinSegs sg) {
  const int i = blockIdx.x, lane = threadIdx.x;
  float v = 0.f;
  for (int b = lane; b < nb; b += 64) v += part[(size_t)b * stride + i];
  v = wave_sum(v);
  if (lane == 0) {
    int r = i;
    for (int k = 0; k < sg.nseg; ++k) {
      if (r < sg.len[k]) { sg.dst[k][r] += v; return; }
      r -= sg.len[k];
    }
  }
}

// LT > 0: L == LT known at compile time -> the 3*(L*L+L) weight/bias gradient partials live in registers over the
// thread's elements and are reduced once per wave with shuffles (the generic form does one LDS atomic per partial
// and element: 256 threads x 60 atomics on the same 60 addresses per iteration = 93 us for a 131k-element map).
template <int LT>
__global__ __launch_bounds__(256) void latent_bwd_kernel(LatBwdArgs a) {
  extern __shared__ float sm[];  // LT > 0: [4 waves][3*(L*L+L)], else 3*(L*L+L)
  const int L = LT > 0 ? LT : a.L, LL = L * L + L;
  constexpr int NACC = LT > 0 ? 3 * (LT * LT + LT) : 1;
  float racc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) racc[i] = 0.f;
  auto add = [&](float* lds_slot, int ridx, float v) {
    if constexpr (LT > 0) racc[ridx] += v; else atomicAdd(lds_slot, v);
  };
  for (int i = threadIdx.x; i < 3 * LL; i += 256) sm[i] = 0.f;
  __syncthreads();
  float* s_wm = sm; float* s_bm = sm + L * L;
  float* s_wl = sm + LL; float* s_bl = s_wl + L * L;
  float* s_wp = sm + 2 * LL; float* s_bp = s_wp + L * L;
  const long long total = (long long)a.B * a.HW;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int p = e % a.HW;
    const int b = e / a.HW;
    float hv[MAXL], z[MAXL], sg[MAXL], ep[MAXL], dz[MAXL], dm[MAXL], dl[MAXL];
    bool inr[MAXL];
    for (int j = 0; j < L; ++j) hv[j] = a.h[e * L + j];
    for (int i = 0; i < L; ++i) {
      float m = a.bm[i], l = a.bl[i];
      for (int j = 0; j < L; ++j) {
        m += a.wm[i * L + j] * hv[j];
        l += a.wl[i * L + j] * hv[j];
      }
      inr[i] = (l >= -30.f) && (l <= 20.f);   // torch.clamp passes the gradient on the closed range
      l = fminf(fmaxf(l, -30.f), 20.f);
      sg[i] = expf(0.5f * l);
      const size_t o = ((size_t)b * L + i) * a.HW + p;
      ep[i] = a.eps ? a.eps[o] : 0.f;
      z[i] = m + ep[i] * sg[i];
    }
    for (int j = 0; j < L; ++j) {
      float v = 0.f;
      if (a.dzq)
        for (int i = 0; i < L; ++i) v += a.wp[i * L + j] * a.dzq[e * L + i];
      dz[j] = v;
    }
    if (a.dzq) {
      for (int i = 0; i < L; ++i) {
        const float g = a.dzq[e * L + i];
        add(&s_bp[i], 2 * LL + L * L + i, g);
        for (int j = 0; j < L; ++j) add(&s_wp[i * L + j], 2 * LL + i * L + j, g * z[j]);
      }
    }
    for (int i = 0; i < L; ++i) {
      const size_t o = ((size_t)b * L + i) * a.HW + p;
      dm[i] = dz[i] + (a.dmu ? a.dmu[o] : 0.f);
      const float dsg = dz[i] * ep[i] + (a.dsigma ? a.dsigma[o] : 0.f);
      dl[i] = inr[i] ? dsg * sg[i] * 0.5f : 0.f;
      add(&s_bm[i], L * L + i, dm[i]);
      add(&s_bl[i], LL + L * L + i, dl[i]);
      for (int j = 0; j < L; ++j) {
        add(&s_wm[i * L + j], i * L + j, dm[i] * hv[j]);
        add(&s_wl[i * L + j], LL + i * L + j, dl[i] * hv[j]);
      }
    }
    for (int j = 0; j < L; ++j) {
      float v = 0.f;
      for (int i = 0; i < L; ++i) v += a.wm[i * L + j] * dm[i] + a.wl[i * L + j] * dl[i];
      a.dh[e * L + j] = v;
    }
  }
  if constexpr (LT > 0) {   // a wave's LDS row is laid out exactly like racc: [wm | bm | wl | bl | wp | bp]
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      const float v = wave_sum(racc[i]);
      if ((threadIdx.x & 63) == 0) sm[(threadIdx.x >> 6) * NACC + i] = v;
    }
  }
  __syncthreads();
  // block partial row (plain stores); block_partials_finalize_kernel adds the rows up in a fixed order
  for (int i = threadIdx.x; i < 3 * LL; i += 256) {
    float v = sm[i];
    if constexpr (LT > 0) v = (sm[i] + sm[NACC + i]) + (sm[2 * NACC + i] + sm[3 * NACC + i]);
    a.part[(size_t)blockIdx.x * 3 * LL + i] = v;
  }
}

// Generic latent width: same two-phase scheme as post_quant_bwd_generic_kernel for the three L x L parameter gradients
// (gwm = sum dm (x) h, gwl = sum dl (x) h, gwp = sum dzq (x) z) and their bias gradients.
__global__ __launch_bounds__(256) void latent_bwd_generic_kernel(LatBwdArgs a) {
  extern __shared__ float sm[];   // five [GCH][L] tables: dm, dl, h, dzq, z
  const int L = a.L, LL = L * L + L;
  float* s_dm = sm;
  float* s_dl = sm + GCH * L;
  float* s_hv = sm + 2 * GCH * L;
  float* s_g = sm + 3 * GCH * L;
  float* s_z = sm + 4 * GCH * L;
  const int t = threadIdx.x;
  const int pi = t / L, pj = t - pi * L;
  const bool pair = t < L * L;
  float awm = 0.f, awl = 0.f, awp = 0.f, abm = 0.f, abl = 0.f, abp = 0.f;
  const long long total = (long long)a.B * a.HW, nch = (total + GCH - 1) / GCH;
  for (long long c = blockIdx.x; c < nch; c += gridDim.x) {
    const long long e = c * GCH + t;
    if (t < GCH) {
      if (e < total) {
        const int p = e % a.HW;
        const int b = e / a.HW;
        float hv[MAXL], sg[MAXL], ep[MAXL], dzv[MAXL], dm[MAXL], dl[MAXL], gq[MAXL];
        bool inr[MAXL];
        for (int j = 0; j < L; ++j) hv[j] = a.h[e * L + j];
        for (int i = 0; i < L; ++i) {
          float m = a.bm[i], l = a.bl[i];
          for (int j = 0; j < L; ++j) {
            m += a.wm[i * L + j] * hv[j];
            l += a.wl[i * L + j] * hv[j];
          }
          inr[i] = (l >= -30.f) && (l <= 20.f);   // torch.clamp passes the gradient on the closed range
          l = fminf(fmaxf(l, -30.f), 20.f);
          sg[i] = expf(0.5f * l);
          const size_t o = ((size_t)b * L + i) * a.HW + p;
          ep[i] = a.eps ? a.eps[o] : 0.f;
          s_z[t * L + i] = m + ep[i] * sg[i];
          gq[i] = a.dzq ? a.dzq[e * L + i] : 0.f;
          s_g[t * L + i] = gq[i];
          s_hv[t * L + i] = hv[i];
        }
        for (int j = 0; j < L; ++j) {
          float v = 0.f;
          for (int i = 0; i < L; ++i) v += a.wp[i * L + j] * gq[i];
          dzv[j] = v;
        }
        for (int i = 0; i < L; ++i) {
          const size_t o = ((size_t)b * L + i) * a.HW + p;
          dm[i] = dzv[i] + (a.dmu ? a.dmu[o] : 0.f);
          const float dsg = dzv[i] * ep[i] + (a.dsigma ? a.dsigma[o] : 0.f);
          dl[i] = inr[i] ? dsg * sg[i] * 0.5f : 0.f;
          s_dm[t * L + i] = dm[i];
          s_dl[t * L + i] = dl[i];
        }
        for (int j = 0; j < L; ++j) {
          float v = 0.f;
          for (int i = 0; i < L; ++i) v += a.wm[i * L + j] * dm[i] + a.wl[i * L + j] * dl[i];
          a.dh[e * L + j] = v;
        }
      } else {
        for (int i = 0; i < L; ++i)
          s_dm[t * L + i] = s_dl[t * L + i] = s_hv[t * L + i] = s_g[t * L + i] = s_z[t * L + i] = 0.f;
      }
    }
    __syncthreads();
    if (pair)
      for (int k = 0; k < GCH; ++k) {
        const float h_ = s_hv[k * L + pj];
        awm += s_dm[k * L + pi] * h_;
        awl += s_dl[k * L + pi] * h_;
        awp += s_g[k * L + pi] * s_z[k * L + pj];
      }
    if (t < L)
      for (int k = 0; k < GCH; ++k) {
        abm += s_dm[k * L + t];
        abl += s_dl[k * L + t];
        abp += s_g[k * L + t];
      }
    __syncthreads();
  }
  // block partial row, laid out [wm | bm | wl | bl | wp | bp] like the specialised kernel's
  float* row = a.part + (size_t)blockIdx.x * 3 * LL;
  if (pair) {
    row[pi * L + pj] = awm;
    row[LL + pi * L + pj] = awl;
    row[2 * LL + pi * L + pj] = awp;
  }
  if (t < L) {
    row[L * L + t] = abm;
    row[LL + L * L + t] = abl;
    row[2 * LL + L * L + t] = abp;
  }
}

// ---- loss -------------------------------------------------------------------------------------
// part[block] = {sum |r-x| or (r-x)^2, sum_kl} of the block's elements (out2 += their fixed-order total); gradient seeds written scaled so that
// d(total)/d(.) with total = mean_recon + kl_weight * mean_b(kl)  (losses.py:62-66 with the other
// weights zero).  third_mode 0: third is used as log-variance (the reference call, train_vae.py:394);
// 1: third is sigma with input_is_logvar=False (losses.py:25-26).
__global__ __launch_bounds__(256) void vae_loss_kernel(const float* __restrict__ recon, const float* __restrict__ img,
                                                       long long npix, const float* __restrict__ mu,
                                                       const float* __restrict__ third, long long nlat, float* part,
                                                       float* d_recon, float* d_mu, float* d_third, int l2,
                                                       int third_mode, float kl_weight, float inv_npix, float inv_b) {
  float sr = 0.f, sk = 0.f;
  const long long stride = (long long)gridDim.x * 256;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < npix; e += stride) {
    const float d = recon[e] - img[e];
    if (l2) {
      sr += d * d;
      if (d_recon) d_recon[e] = 2.f * d * inv_npix;
    } else {
      sr += fabsf(d);
      if (d_recon) d_recon[e] = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * inv_npix;
    }
  }
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < nlat; e += stride) {
    const float m = mu[e], t = third[e];
    float s = t, dsdt = 1.f;
    if (third_mode == 1) {
      s = logf(t * t + 1e-8f);
      dsdt = 2.f * t / (t * t + 1e-8f);
    }
    const float es = expf(s);
    sk += -0.5f * (1.f + s - m * m - es);
    if (d_mu) d_mu[e] = kl_weight * inv_b * m;
    if (d_third) d_third[e] = kl_weight * inv_b * (-0.5f * (1.f - es)) * dsdt;
  }
  sr = wave_sum(sr);
  sk = wave_sum(sk);
  __shared__ float red[8];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[wave] = sr; red[4 + wave] = sk; }
  __syncthreads();
  if (threadIdx.x == 0) {   // block partial; block_partials_finalize_kernel adds the blocks up in a fixed order
    part[2 * blockIdx.x] = ((red[0] + red[1]) + (red[2] + red[3])) * inv_npix;
    part[2 * blockIdx.x + 1] = ((red[4] + red[5]) + (red[6] + red[7])) * inv_b;
  }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long long n,
                                                   float lr, float b1, float b2, float eps, float bc1, float bc2_sqrt,
                                                   float grad_scale) {
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    const float gr = g[e] * grad_scale;
    const float mm = b1 * m[e] + (1.f - b1) * gr;
    const float vv = b2 * v[e] + (1.f - b2) * gr * gr;
    m[e] = mm;
    v[e] = vv;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    p[e] -= (lr / bc1) * (mm / denom);
  }
}

__global__ __launch_bounds__(256) void cast_nchw_f32_to_nhwc_bf16(const float* __restrict__ x, bf16* __restrict__ y,
                                                                  int N, int C, int HW) {
  const long long total = (long long)N * C * HW;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int c = e % C;
    const long long t = e / C;
    const int p = t % HW;
    const int n = t / HW;
    y[e] = (bf16)x[((size_t)n * C + c) * HW + p];
  }
}
__global__ __launch_bounds__(256) void cast_nhwc_bf16_to_nchw_f32(const bf16* __restrict__ x, float* __restrict__ y,
                                                                  int N, int C, int HW) {
  const long long total = (long long)N * C * HW;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int p = e % HW;
    const long long t = e / HW;
    const int c = t % C;
    const int n = t / C;
    y[e] = (float)x[((size_t)n * HW + p) * C + c];
  }
}

// ---- AR-VAE attribute regularisation (reference src/pti_ldm_vae/models/losses.py:69-166; call site ------------------
// vae_scripts/train_vae.py:403-417).  z = z_mu.mean(h, w); for every attribute mapped to latent channel ch with slope
// delta: over the ordered pairs (i, j), i != j, of the local batch (optionally a sampled subset, given as a byte mask)
// whose attribute values differ, mean of (tanh(delta * (z_j - z_i)) - sign(a_j - a_i))^2.  One workgroup per latent
// channel (it walks the attributes mapped to that channel in order, so two attributes on one channel never race):
// wave-per-sample spatial means, thread-per-sample row/column sums over the b x b pair grid (every sum in a fixed
// order: no atomics, bitwise reproducible), and the gradient gamma * d(sum of the per-attribute losses)/d z_mu is
// ADDED to d_mu (the same value at each of the hw positions of (sample, ch): d mean / d element = 1 / hw).
constexpr int AR_MAXB = 1024;
struct ArArgs {
  const float* mu;        // [b][l][hw]
  const float* attrs;     // [na][b]
  const int* channels;    // [na]
  const float* deltas;    // [na]
  const unsigned char* mask;   // [na][b][b] or nullptr (= every ordered pair)
  float* per_attr;        // [na]
  int* counts;            // [na]
  float* d_mu;            // [b][l][hw] or nullptr
  int b, l, hw, na;
  float gamma;
};

__device__ __forceinline__ float block_sum_256(float v, float* red) {   // fixed order: wave shuffles, then 4 partials
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void ar_vae_kernel(ArArgs a) {
  __shared__ float z[AR_MAXB], at[AR_MAXB], dz[AR_MAXB], red[4];
  const int ch = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  bool mine = false;
  for (int q = 0; q < a.na; ++q) mine |= (a.channels[q] == ch);
  if (!mine) return;   // block-uniform
  const float inv_hw = 1.0f / (float)a.hw;
  for (int s = wave; s < a.b; s += 4) {   // spatial mean of (sample s, channel ch)
    const float* p = a.mu + ((size_t)s * a.l + ch) * a.hw;
    float acc = 0.f;
    for (int i = lane; i < a.hw; i += 64) acc += p[i];
    acc = wave_sum(acc);
    if (lane == 0) z[s] = acc * inv_hw;
  }
  for (int s = tid; s < a.b; s += 256) dz[s] = 0.f;
  __syncthreads();
  for (int q = 0; q < a.na; ++q) {
    if (a.channels[q] != ch) continue;
    const float delta = a.deltas[q];
    const unsigned char* m = a.mask ? a.mask + (size_t)q * a.b * a.b : nullptr;
    for (int s = tid; s < a.b; s += 256) at[s] = a.attrs[(size_t)q * a.b + s];
    __syncthreads();
    // pass 1: loss numerator and pair count, row k per thread
    float num = 0.f, cnt = 0.f;
    for (int k = tid; k < a.b; k += 256) {
      const float zk = z[k], ak = at[k];
      for (int j = 0; j < a.b; ++j) {
        const float d = at[j] - ak;
        const float order = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        const bool sel = order != 0.f && j != k && (!m || m[(size_t)k * a.b + j]);
        if (sel) {
          const float e = tanhf(delta * (z[j] - zk)) - order;
          num += e * e;
          cnt += 1.f;
        }
      }
    }
    num = block_sum_256(num, red);
    cnt = block_sum_256(cnt, red);
    if (tid == 0) {
      a.per_attr[q] = cnt > 0.f ? num / cnt : 0.f;
      a.counts[q] = (int)cnt;
    }
    // pass 2: d loss_q / d z_k = sum_i G[i][k] - sum_j G[k][j],  G[i][j] = 2 (pred - order) sel delta (1 - pred^2) / cnt
    if (a.d_mu && cnt > 0.f) {
      const float sc = 2.0f * delta / cnt;
      for (int k = tid; k < a.b; k += 256) {
        const float zk = z[k], ak = at[k];
        float g = 0.f;
        for (int i = 0; i < a.b; ++i) {
          if (i == k) continue;
          const float d = at[i] - ak;                       // a_i - a_k
          const float o_ki = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);   // order[k][i] = sign(a_i - a_k); order[i][k] = -o_ki
          if (o_ki == 0.f) continue;
          const float p_ki = tanhf(delta * (z[i] - zk));     // pred[k][i]; pred[i][k] = -p_ki
          const float w = (p_ki - o_ki) * (1.f - p_ki * p_ki);
          // G[i][k] = sc * (pred[i][k] - order[i][k]) (1 - pred^2) = -sc * w ; G[k][i] = sc * w
          if (!m || m[(size_t)i * a.b + k]) g -= sc * w;     // + G[i][k]
          if (!m || m[(size_t)k * a.b + i]) g -= sc * w;     // - G[k][i]
        }
        dz[k] += g;
      }
    }
    __syncthreads();
  }
  if (a.d_mu) {
    const float sc = a.gamma * inv_hw;
    for (int s = 0; s < a.b; ++s) {
      float* p = a.d_mu + ((size_t)s * a.l + ch) * a.hw;
      const float g = dz[s] * sc;
      for (int i = tid; i < a.hw; i += 256) p[i] += g;
    }
  }
}

inline unsigned nblocks(long long total, int cap = 4096) {
  long long b = (total + 255) / 256;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

extern "C" int pti_latent_head_fwd(const float* h, const float* eps, const float* wm, const float* bm,
                                   const float* wl, const float* bl, const float* wp, const float* bp, float* mu,
                                   float* sigma, float* logvar, float* zq, int b, int hw, int l, pti_stream_t s) {
  if (!h || !wm || !bm || !wl || !bl || !wp || !bp || !mu || !sigma || !zq) PTI_FAIL(PTI_EINVAL, "latent_head_fwd: null pointer");
  if (l <= 0 || l > MAXL || b <= 0 || hw <= 0) PTI_FAIL(PTI_EUNSUPPORTED, "latent_head_fwd: latent channels %d (max %d)", l, MAXL);
  LatArgs a{h, eps, wm, bm, wl, bl, wp, bp, mu, sigma, logvar, zq, b, hw, l};
  PTI_LAUNCH(latent_fwd_kernel, dim3(nblocks((long long)b * hw)), dim3(256), 0, (hipStream_t)s, a);
  PTI_CHECK_LAUNCH("latent_head_fwd");
  return PTI_OK;
}

extern "C" int pti_post_quant(const float* z_nchw, const float* wp, const float* bp, float* zq_nhwc, int b, int hw,
                              int l, pti_stream_t s) {
  if (!z_nchw || !wp || !bp || !zq_nhwc || l <= 0 || l > MAXL) PTI_FAIL(PTI_EINVAL, "post_quant: bad args");
  PTI_LAUNCH(post_quant_kernel, dim3(nblocks((long long)b * hw)), dim3(256), 0, (hipStream_t)s, z_nchw, wp, bp,
                     zq_nhwc, b, hw, l);
  PTI_CHECK_LAUNCH("post_quant");
  return PTI_OK;
}

extern "C" int pti_post_quant_bwd(const float* dzq_nhwc, const float* z_nchw, const float* wp, float* dz_nchw,
                                  float* gwp, float* gbp, float* workspace, int b, int hw, int l, pti_stream_t s) {
  if (!dzq_nhwc || !z_nchw || !wp || !gwp || !gbp || !workspace || l <= 0 || l > MAXL) PTI_FAIL(PTI_EINVAL, "post_quant_bwd: bad args");
  const unsigned nb = nblocks((long long)b * hw, PTI_POST_QUANT_BWD_MAX_BLOCKS);
  const int ll = l * l + l;
  if (l == 4)
    PTI_LAUNCH(post_quant_bwd_kernel<4>, dim3(nb), dim3(256), 4 * ll * sizeof(float), (hipStream_t)s, dzq_nhwc,
                       z_nchw, wp, dz_nchw, workspace, b, hw, l);
  else   // other latent widths: two-phase kernel, fixed summation order
    PTI_LAUNCH(post_quant_bwd_generic_kernel, dim3(nb), dim3(256), 2 * GCH * l * sizeof(float), (hipStream_t)s, dzq_nhwc, z_nchw,
                       wp, dz_nchw, workspace, b, hw, l);
  PTI_CHECK_LAUNCH("post_quant_bwd");
  FinSegs sg{{gwp, gbp}, {l * l, l}, 2};
  PTI_LAUNCH(block_partials_finalize_kernel, dim3(ll), dim3(64), 0, (hipStream_t)s, workspace, (int)nb, ll, sg);
  PTI_CHECK_LAUNCH("post_quant_bwd_finalize");
  return PTI_OK;
}

extern "C" int pti_latent_head_bwd(const float* h, const float* eps, const float* wm, const float* bm,
                                   const float* wl, const float* bl, const float* wp, const float* bp,
                                   const float* dzq, const float* dmu, const float* dsigma, float* dh, float* gwm,
                                   float* gbm, float* gwl, float* gbl, float* gwp, float* gbp, float* workspace,
                                   int b, int hw, int l, pti_stream_t s) {
  if (!h || !wm || !bm || !wl || !bl || !wp || !bp || !dh || !gwm || !gbm || !gwl || !gbl || !gwp || !gbp || !workspace)
    PTI_FAIL(PTI_EINVAL, "latent_head_bwd: null pointer");
  if (l <= 0 || l > MAXL) PTI_FAIL(PTI_EUNSUPPORTED, "latent_head_bwd: latent channels %d", l);
  LatBwdArgs a{h, eps, wm, bm, wl, bl, wp, bp, dzq, dmu, dsigma, dh, gwm, gbm, gwl, gbl, gwp, gbp, b, hw, l, workspace};
  const int ll = l * l + l;
  long long nb;
  if (l == 4)
  {   // ~4 elements per thread: the 60 wave reductions at the end are amortised
    nb = ((long long)b * hw + 1023) / 1024;
    nb = nb < 1 ? 1 : (nb > PTI_LATENT_BWD_MAX_BLOCKS ? PTI_LATENT_BWD_MAX_BLOCKS : nb);
    PTI_LAUNCH(latent_bwd_kernel<4>, dim3((unsigned)nb), dim3(256), 4 * 3 * ll * sizeof(float), (hipStream_t)s, a);
  }
  else {
    // other latent widths: two-phase kernel (one chunk of 128 elements per block and iteration), fixed summation order
    nb = ((long long)b * hw + GCH - 1) / GCH;
    nb = nb < 1 ? 1 : (nb > PTI_LATENT_BWD_MAX_BLOCKS ? PTI_LATENT_BWD_MAX_BLOCKS : nb);
    PTI_LAUNCH(latent_bwd_generic_kernel, dim3((unsigned)nb), dim3(256), 5 * GCH * l * sizeof(float), (hipStream_t)s, a);
  }
  PTI_CHECK_LAUNCH("latent_head_bwd");
  FinSegs sg{{gwm, gbm, gwl, gbl, gwp, gbp}, {l * l, l, l * l, l, l * l, l}, 6};
  PTI_LAUNCH(block_partials_finalize_kernel, dim3(3 * ll), dim3(64), 0, (hipStream_t)s, workspace, (int)nb, 3 * ll, sg);
  PTI_CHECK_LAUNCH("latent_head_bwd_finalize");
  return PTI_OK;
}

extern "C" int pti_vae_loss(const float* recon, const float* images, int64_t npix, const float* mu,
                            const float* third, int64_t nlat, int batch, float* out2, float* d_recon, float* d_mu,
                            float* d_third, float* workspace, int l2, int third_mode, float kl_weight, pti_stream_t s) {
  if (!recon || !images || !mu || !third || !out2 || !workspace || npix <= 0 || nlat <= 0 || batch <= 0)
    PTI_FAIL(PTI_EINVAL, "vae_loss: bad args");
  const unsigned nb = nblocks(npix, PTI_VAE_LOSS_MAX_BLOCKS);
  PTI_LAUNCH(vae_loss_kernel, dim3(nb), dim3(256), 0, (hipStream_t)s, recon, images,
                     (long long)npix, mu, third, (long long)nlat, workspace, d_recon, d_mu, d_third, l2, third_mode, kl_weight,
                     1.0f / (float)npix, 1.0f / (float)batch);
  PTI_CHECK_LAUNCH("vae_loss");
  FinSegs sg{{out2}, {2}, 1};
  PTI_LAUNCH(block_partials_finalize_kernel, dim3(2), dim3(64), 0, (hipStream_t)s, workspace, (int)nb, 2, sg);
  PTI_CHECK_LAUNCH("vae_loss_finalize");
  return PTI_OK;
}

extern "C" int pti_ar_vae_loss(const float* mu_nchw, int b, int l, int hw, const float* attrs, const int32_t* channels,
                               const float* deltas, int na, const uint8_t* pair_mask, float gamma, float* per_attr,
                               int32_t* counts, float* d_mu, pti_stream_t s) {
  if (!mu_nchw || !attrs || !channels || !deltas || !per_attr || !counts) PTI_FAIL(PTI_EINVAL, "ar_vae_loss: null pointer");
  if (b <= 0 || l <= 0 || hw <= 0 || na <= 0) PTI_FAIL(PTI_EINVAL, "ar_vae_loss: bad sizes");
  if (b > AR_MAXB) PTI_FAIL(PTI_EUNSUPPORTED, "ar_vae_loss: local batch %d above %d", b, AR_MAXB);
  ArArgs a{mu_nchw, attrs, channels, deltas, pair_mask, per_attr, counts, d_mu, b, l, hw, na, gamma};
  PTI_LAUNCH(ar_vae_kernel, dim3(l), dim3(256), 0, (hipStream_t)s, a);
  PTI_CHECK_LAUNCH("ar_vae_loss");
  return PTI_OK;
}

extern "C" int pti_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                             float beta2, float eps, int step, float grad_scale, pti_stream_t s) {
  if (!p || !g || !m || !v || n <= 0 || step < 1) PTI_FAIL(PTI_EINVAL, "adam_step: bad args");
  // bias corrections in double, as torch.optim.Adam computes them (fp32 powf left ~1e-5 relative error at small steps)
  const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  const float bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  PTI_LAUNCH(adam_kernel, dim3(nblocks(n, 2048)), dim3(256), 0, (hipStream_t)s, p, g, m, v, (long long)n, lr, beta1,
                     beta2, eps, bc1, bc2s, grad_scale);
  PTI_CHECK_LAUNCH("adam_step");
  return PTI_OK;
}

extern "C" int pti_cast_nchw_f32_to_nhwc_bf16(const float* x, void* y, int n, int c, int hw, pti_stream_t s) {
  if (!x || !y) PTI_FAIL(PTI_EINVAL, "cast: null");
  PTI_LAUNCH(cast_nchw_f32_to_nhwc_bf16, dim3(nblocks((long long)n * c * hw)), dim3(256), 0, (hipStream_t)s, x, (bf16*)y, n, c, hw);
  PTI_CHECK_LAUNCH("cast_nchw_f32_to_nhwc_bf16");
  return PTI_OK;
}
extern "C" int pti_cast_nhwc_bf16_to_nchw_f32(const void* x, float* y, int n, int c, int hw, pti_stream_t s) {
  if (!x || !y) PTI_FAIL(PTI_EINVAL, "cast: null");
  PTI_LAUNCH(cast_nhwc_bf16_to_nchw_f32, dim3(nblocks((long long)n * c * hw)), dim3(256), 0, (hipStream_t)s, (const bf16*)x, y, n, c, hw);
  PTI_CHECK_LAUNCH("cast_nhwc_bf16_to_nchw_f32");
  return PTI_OK;
}
