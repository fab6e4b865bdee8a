// Shared device/host helpers for the gfx950 (CDNA4) kernels of libpti_vae_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/pti_vae.h"

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned bu32x4 __attribute__((__vector_size__(16)));   // operand type of the raw_buffer_{load,store}_b128 builtins
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// ---- error plumbing (host) ------------------------------------------------------------------
void pti_set_error(const char* fmt, ...);
#define PTI_FAIL(code, ...)      \
  do {                           \
    pti_set_error(__VA_ARGS__);  \
    return (code);               \
  } while (0)
// Every launch of this library goes through PTI_LAUNCH: it remembers the host-side function pointer of the kernel it
// launched (per thread), so that pti_last_kernel_name() can ask the HIP runtime for the symbol that actually ran -- the
// bench pairs its live timings with rocprofv3 records by that name instead of re-deriving the template arguments.
extern thread_local const void* pti_last_kernel;
#define PTI_LAUNCH(kernel, grid, block, shmem, stream, ...)                     \
  do {                                                                           \
    pti_last_kernel = reinterpret_cast<const void*>(kernel);                     \
    hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);         \
  } while (0)
#define PTI_CHECK_LAUNCH(name)                                             \
  do {                                                                     \
    hipError_t e__ = hipGetLastError();                                    \
    if (e__ != hipSuccess)                                                 \
      PTI_FAIL(PTI_ELAUNCH, "%s: launch failed: %s", name, hipGetErrorString(e__)); \
  } while (0)

// ---- device helpers -------------------------------------------------------------------------
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t lo16) { return __uint_as_float(lo16 << 16); }

// unpack 8 bf16 (one 16-byte piece) to 8 floats
__device__ __forceinline__ void unpack8(const u32x4& r, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(r[i] << 16);
    f[2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u);
  }
}
// pack 8 floats to 8 bf16 (round to nearest even via the hardware cvt)
__device__ __forceinline__ u32x4 pack8(const float* f) {
  u32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    bf16x2 p;
    p[0] = (bf16)f[2 * i];
    p[1] = (bf16)f[2 * i + 1];
    r[i] = __builtin_bit_cast(uint32_t, p);
  }
  return r;
}
__device__ __forceinline__ u32x2 pack4(float a, float b, float c, float d) {
  bf16x2 p0, p1;
  p0[0] = (bf16)a; p0[1] = (bf16)b; p1[0] = (bf16)c; p1[1] = (bf16)d;
  u32x2 r;
  r[0] = __builtin_bit_cast(uint32_t, p0);
  r[1] = __builtin_bit_cast(uint32_t, p1);
  return r;
}
// ---- GroupNorm statistics ---------------------------------------------------------------------------
// {sum, sum of squares} per (sample, group) are accumulated as 64-bit FIXED-POINT integers (Q47.16): integer adds are
// associative, so the totals -- and with them every kernel that normalises by them -- are bitwise reproducible no
// matter in which order workgroups and waves arrive.  (Float atomics made two runs on identical inputs differ by the
// full bf16 rounding noise, 1.6e-2 relative on the reconstruction: the low bits of a mean flip roundings downstream.)
// Per-lane / per-wave partial sums stay fp32 in a fixed order; they are quantised to 2^-16 once, when they enter the
// shared accumulator.  Range: |sum of squares| < 2^47 = 1.4e14.
typedef long long stat_t;
__device__ __forceinline__ unsigned long long stat_q(float v) { return (unsigned long long)__float2ll_rn(v * 65536.0f); }
// value / 2^16 as fp32 from the two 32-bit halves (4 VALU ops; the int64 -> double -> float route cost ~3x as many
// and showed up as +20 % on the VALU-bound 32-channel convs)
__device__ __forceinline__ float stat_f(const stat_t* p, int i) {
  const long long v = p[i];
  return (float)(int)(v >> 32) * 65536.0f + (float)(unsigned)v * (1.0f / 65536.0f);
}
__device__ __forceinline__ void stat_add(stat_t* p, float v) { atomicAdd((unsigned long long*)p, stat_q(v)); }

// ---- 16-bit activation formats -----------------------------------------------------------------
// Activations written by the FORWARD pass may be stored as IEEE fp16 (11-bit significand) instead of bf16: same
// bytes, 8x smaller rounding step, and GroupNorm keeps their range far inside fp16's.  MFMA operands, saved
// activated inputs, attention tensors and every gradient stay bf16.  `f16` flags are wave-uniform.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void unpack2f(uint32_t w, bool f16, float& a, float& b) {
  if (f16) {
    const f16x2 h = __builtin_bit_cast(f16x2, w);
    a = (float)h[0];
    b = (float)h[1];
  } else {
    a = __uint_as_float(w << 16);
    b = __uint_as_float(w & 0xffff0000u);
  }
}
__device__ __forceinline__ uint32_t pack2f(float a, float b, bool f16) {
  if (f16) {
    f16x2 h;
    h[0] = (_Float16)a;
    h[1] = (_Float16)b;
    return __builtin_bit_cast(uint32_t, h);
  }
  bf16x2 p;
  p[0] = (bf16)a;
  p[1] = (bf16)b;
  return __builtin_bit_cast(uint32_t, p);
}
__device__ __forceinline__ void unpack8f(const u32x4& r, float* f, bool f16) {
#pragma unroll
  for (int i = 0; i < 4; ++i) unpack2f(r[i], f16, f[2 * i], f[2 * i + 1]);
}
__device__ __forceinline__ u32x4 pack8f(const float* f, bool f16) {
  u32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = pack2f(f[2 * i], f[2 * i + 1], f16);
  return r;
}
__device__ __forceinline__ u32x2 pack4f(float a, float b, float c, float d, bool f16) {
  u32x2 r;
  r[0] = pack2f(a, b, f16);
  r[1] = pack2f(c, d, f16);
  return r;
}

// MODE.FP16_OVFL (HW_REG_MODE bit 23): an fp32 -> fp16 conversion that overflows gives +-65504 instead of +-inf (a true
// infinity stays one).  Kernels that store fp16 activations switch it on first (one scalar instruction; the mode is per
// wave and dies with it), so an un-normalised residual stream that leaves fp16's range saturates instead of turning
// into inf and then NaN through the next GroupNorm's statistics.  The argument must be wave-uniform.
__device__ __forceinline__ void fp16_saturate_on() { __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 1); }

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + __expf(-v)); }
// d silu(v)/dv = s*(1 + v*(1-s)), s = sigmoid(v)
__device__ __forceinline__ float dsilu_f(float v) {
  float s = 1.0f / (1.0f + __expf(-v));
  return s * (1.0f + v * (1.0f - s));
}

// GroupNorm scale / shift / mean / rstd of NCH consecutive channels ch0 .. ch0+NCH-1 (ch0 % NCH == 0) from the float
// {sum, sum of squares} table `sfl` of this sample.  For power-of-two group sizes (every reference config) the group
// index is a shift and the per-group work -- table read, mean, v_rsq_f32 -- is done once per GROUP on a straight-line
// path per group size; an integer division by a run-time value (~25 VALU instructions) and the denormal-guarded
// rsqrtf per CHANNEL cost the fused GroupNorm-backward epilogue ~500 VALU instructions per thread and tile.
template <int NCH, int GS>
__device__ __forceinline__ void gn_params_groups(const float* sfl, const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, int ch0, int g0, float inv_cnt, float eps,
                                                 float* sc, float* sh, float* mu, float* rs) {
#pragma unroll
  for (int k = 0; k < NCH / GS; ++k) {
    const f32x2 st = *(const f32x2*)(sfl + 2 * (g0 + k));
    const float mean = st[0] * inv_cnt;
    const float rstd = __builtin_amdgcn_rsqf(fmaxf(st[1] * inv_cnt - mean * mean, 0.f) + eps);   // var + eps >= eps: no denormals
#pragma unroll
    for (int jj = 0; jj < GS; ++jj) {
      const int j = k * GS + jj;
      sc[j] = rstd * gamma[ch0 + j];
      sh[j] = beta[ch0 + j] - mean * sc[j];
      mu[j] = mean;
      rs[j] = rstd;
    }
  }
}
template <int NCH>
__device__ __forceinline__ void gn_params(const float* sfl, const float* __restrict__ gamma, const float* __restrict__ beta,
                                          int ch0, int cpg, float inv_cnt, float eps, float* sc, float* sh, float* mu,
                                          float* rs) {
  if ((cpg & (cpg - 1)) == 0) {   // wave-uniform
    const int g0 = ch0 >> __builtin_ctz(cpg);
    if (cpg >= NCH) gn_params_groups<NCH, NCH>(sfl, gamma, beta, ch0, g0, inv_cnt, eps, sc, sh, mu, rs);
    else if (NCH > 4 && cpg == 4) gn_params_groups<NCH, (NCH > 4 ? 4 : NCH)>(sfl, gamma, beta, ch0, g0, inv_cnt, eps, sc, sh, mu, rs);
    else if (cpg == 2) gn_params_groups<NCH, 2>(sfl, gamma, beta, ch0, g0, inv_cnt, eps, sc, sh, mu, rs);
    else gn_params_groups<NCH, 1>(sfl, gamma, beta, ch0, g0, inv_cnt, eps, sc, sh, mu, rs);
  } else {
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int g = (ch0 + j) / cpg;
      const float mean = sfl[2 * g] * inv_cnt;
      const float rstd = __builtin_amdgcn_rsqf(fmaxf(sfl[2 * g + 1] * inv_cnt - mean * mean, 0.f) + eps);
      sc[j] = rstd * gamma[ch0 + j];
      sh[j] = beta[ch0 + j] - mean * sc[j];
      mu[j] = mean;
      rs[j] = rstd;
    }
  }
}

// SiLU(x * sc + sh) of two elements with the packed-fp32 VALU ops (v_pk_fma / v_pk_mul / v_pk_add: two lanes-worth per
// issue) around the two transcendentals each element needs (v_exp_f32, v_rcp_f32): 4 packed + 4 scalar issues per
// PAIR instead of 6 scalar issues per element.  The narrow-layer convs are VALU-bound in their GroupNorm+SiLU loader.
__device__ __forceinline__ f32x2 gn_silu2(f32x2 x, f32x2 sc, f32x2 sh) {
  const f32x2 v = x * sc + sh;
  const f32x2 t = v * -1.4426950408889634f;   // exp(-v) = 2^(-v * log2 e)
  f32x2 e;
  e[0] = __builtin_amdgcn_exp2f(t[0]);
  e[1] = __builtin_amdgcn_exp2f(t[1]);
  const f32x2 d = e + 1.0f;
  f32x2 r;
  r[0] = __builtin_amdgcn_rcpf(d[0]);
  r[1] = __builtin_amdgcn_rcpf(d[1]);
  return v * r;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
