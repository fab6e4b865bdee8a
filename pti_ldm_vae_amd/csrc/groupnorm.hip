// GroupNorm statistics and backward for NHWC bf16 tensors (gfx950).
//
// Replaces the reduction half of nn.GroupNorm (MONAI AEKLResBlock.norm1/norm2, the attention
// norm and the final Encoder/Decoder norm).  The apply half (+SiLU) is never
// a kernel of its own: it lives in the loader of the consuming convolution (conv_mfma.hip,
// conv_direct.hip).  Purely HBM-bound: every thread streams 16-byte pieces (8 channels).
#include "pti_common.h"

namespace {

// stats[n][g] += {sum, sumsq}.  grid = (blocks_per_sample, N), block = 256.
__global__ __launch_bounds__(256) void gn_stats_kernel(const bf16* __restrict__ x, float* __restrict__ stats,
                                                       int HW, int C, int G, int pix_per_block) {
  extern __shared__ float sm[];  // [G][2]
  const int n = blockIdx.y;
  const int tid = threadIdx.x;
  const int NC = C / 8;
  const int ppi = 256 / NC;  // pixels per iteration (NC divides 256 for C in {32..2048} powers of two)
  const int lc = tid % NC, lp = tid / NC;
  for (int i = tid; i < 2 * G; i += 256) sm[i] = 0.f;
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
      unpack8(r[u], f);
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
    unpack8(r, f);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      s[j] += f[j];
      q[j] += f[j] * f[j];
    }
  }
  const int cpg = C / G;
  if (cpg >= 8) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a += s[j]; b += q[j]; }
    const int g = (lc * 8) / cpg;
    atomicAdd(&sm[2 * g], a);
    atomicAdd(&sm[2 * g + 1], b);
  } else {
    for (int j0 = 0; j0 < 8; j0 += cpg) {
      float a = 0.f, b = 0.f;
      for (int j = j0; j < j0 + cpg; ++j) { a += s[j]; b += q[j]; }
      const int g = (lc * 8 + j0) / cpg;
      atomicAdd(&sm[2 * g], a);
      atomicAdd(&sm[2 * g + 1], b);
    }
  }
  __syncthreads();
  for (int i = tid; i < 2 * G; i += 256) atomicAdd(&stats[(size_t)n * G * 2 + i], sm[i]);
}

}  // namespace

extern "C" int pti_gn_stats(const void* x, float* stats, int n, int hw, int c, int groups, pti_stream_t s) {
  if (!x || !stats || n <= 0 || hw <= 0) PTI_FAIL(PTI_EINVAL, "gn_stats: bad pointer/dims");
  if (c < 8 || c > 2048 || (c & (c - 1)) || groups <= 0 || c % groups)
    PTI_FAIL(PTI_EUNSUPPORTED, "gn_stats: c=%d must be a power of two in [8,2048] and divisible by groups=%d", c, groups);
  const int cpg = c / groups;
  if (cpg < 8 && (8 % cpg)) PTI_FAIL(PTI_EUNSUPPORTED, "gn_stats: channels/group %d", cpg);
  if (cpg >= 8 && (cpg % 8)) PTI_FAIL(PTI_EUNSUPPORTED, "gn_stats: channels/group %d", cpg);
  const int ppi = 256 / (c / 8);
  // aim for >= ~2048 blocks over the whole launch, each block a multiple of ppi pixels
  int bps = cdiv(2048, n);
  int ppb = cdiv(hw, bps);
  ppb = cdiv(ppb, 4 * ppi) * 4 * ppi;
  bps = cdiv(hw, ppb);
  hipLaunchKernelGGL(gn_stats_kernel, dim3(bps, n), dim3(256), 2 * groups * sizeof(float), (hipStream_t)s,
                     (const bf16*)x, stats, hw, c, groups, ppb);
  PTI_CHECK_LAUNCH("gn_stats");
  return PTI_OK;
}
