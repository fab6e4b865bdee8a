// Shared by conv_mfma.hip (v1 / v2 kernels, dispatch, weight packing) and conv_ws.hip (weight-stationary persistent
// kernel for the 128 -> 128 3x3 layers): launch arguments, the MFMA wrapper, the half-wave fold, the v2 tile geometry.
#pragma once
#include "pti_common.h"

namespace pti_conv {

struct ConvArgs {
  const bf16* x;
  const unsigned char* w;
  const float* bias;
  const stat_t* in_stats;
  const float* gamma;
  const float* beta;
  const bf16* res;
  bf16* y;
  stat_t* out_stats;
  int N, H, W, Cin, Ho, Wo, Cout;
  int mode, prologue, groups, out_groups;
  float eps, inv_cnt;
  int tiles_x, tiles_y;
  // fused GroupNorm(+SiLU) backward reduction (data-gradient launches): res = GN input gx (same shape as y);
  // y = dA * act'(GN(gx)) and g_sums[n][c] += {sum dy, sum dy*xhat}
  int gn_mode;            // 0 off, 1 GN, 2 GN+SiLU
  int g_groups;
  float g_inv_cnt, g_eps;
  const stat_t* g_stats; const float* g_gamma; const float* g_beta;
  float* g_sums;          // PARTIAL sums [N][g_T][Cout][2] (one slot per pixel tile, plain stores; pti_gn_sums_finalize adds them up)
  int g_T;                // pixel tiles per sample
  // optional side output (v2 kernel, PTI_CONV_S1 with a prologue): the activated input act(GN(x)) as bf16 NHWC,
  // written by the cout-tile-0 workgroups from their staging registers, for the weight-gradient pass to reuse
  bf16* act_out;
  // 16-bit storage format of x / residual (or the GN input of the fused backward) / y: 0 = bf16, 1 = fp16
  int in_f16, res_f16, out_f16;
  int pool2;   // v2 kernel: store the 2x2-sum-pooled output tile [N][Ho/2][Wo/2][Cout] (data gradient of nearest-2x up-sampling)
  int relu_out;   // y = max(y, 0) in the epilogue (plain fp16 forward launches only: the perceptual network's Fire convs)
  int w_f16;   // packed weights are IEEE fp16 and the MFMA runs v_mfma_f32_32x32x16_f16 on fp16 operands (forward convs on fp16 storage)
  // prologue PRO_GNB (the GroupNorm backward of the layer ABOVE, applied while the input is staged): x = g = dA * act'(GN(x2)),
  // x2 = that GroupNorm's input, in_stats / gamma = its statistics / weight, p_sums = its finalized {sum g, sum g*xhat}
  // per (n, channel); the staged value is  rstd * (gamma * g - c1 - xhat * c2)
  const bf16* x2;
  const float* p_sums;
  int x2_f16;
};
constexpr int PRO_GNB = 3;   // (internal: not a PTI_PRO_* value of the C-ABI's pti_conv_desc)

// One 32x32x16 MFMA step on bf16 or (OPH) fp16 operands; the fragment registers are typed bf16x8 either way.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <bool OPH>
__device__ __forceinline__ f32x16 mfma32(const bf16x8& a, const bf16x8& b, const f32x16& c) {
  if constexpr (OPH)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__host__ __device__ constexpr int pick_ck2(int cin, int ct) {
  const int ck = cin % 128 == 0 ? 128 : (cin % 64 == 0 ? 64 : 32);
  const int cap = ct >= 128 ? 128 : (ct == 64 ? 64 : 32);   // keep the halo tile <= ~48 KiB
  return ck < cap ? ck : cap;
}

// Sum NV per-lane values over the 32 lanes of each wave half with NV-1 (+ log2(32/NV)) shuffles instead of
// 5*NV: at every level a lane keeps one half of its values and trades the other half with its partner.
// Afterwards v[0] of lane j holds the total of value index  (j & 31) >> (5 - log2 NV)  (all lanes of that
// index hold the same total).
template <int NV, int N, int O>
__device__ __forceinline__ void fold32_level(float (&v)[NV], int j) {
  if constexpr (O >= 1) {
    if constexpr (N > 1) {
      const bool hi = (j & O) != 0;
#pragma unroll
      for (int i = 0; i < N / 2; ++i) {
        const float send = hi ? v[i] : v[i + N / 2];
        const float keep = hi ? v[i + N / 2] : v[i];
        v[i] = keep + __shfl_xor(send, O, 64);
      }
    } else {
      v[0] += __shfl_xor(v[0], O, 64);
    }
    fold32_level<NV, (N > 1 ? N / 2 : 1), O / 2>(v, j);
  }
}
template <int NV>
__device__ __forceinline__ void fold32(float (&v)[NV], int j) { fold32_level<NV, NV, 16>(v, j); }

template <int KS, int CK, int CT, int PXF>
struct Cfg2 {
  static constexpr int WN = CT / 32, WM = 4 / WN;
  static constexpr int TH2 = 2 * PXF * WM, TW2 = 16;   // PXF = MFMA pixel fragments (2 rows x 16) per wave
  static constexpr int HH = TH2 + KS - 1, HW = TW2 + KS - 1;
  static constexpr int NP = HH * HW;
  static constexpr int NC = CK / 8;
  static constexpr int PIXB = CK * 2;
  static constexpr int NT = CT / 32;
  static constexpr int KPC = CK / 16;
  static constexpr int KBC = KS * KS * KPC;
  // weight fragments per register set (two sets alternate); divides KBC with an even quotient for 3x3.
  // (tried: R = 3 + 3 workgroups/CU for CK = 32 -> spills, 30 % slower)
  static constexpr int R = (KS == 3) ? (PXF == 2 ? 3 : 9) : KBC;
  static constexpr int WGS_PER_CU = PXF == 2 ? 4 : 2;
  static constexpr int HALO_BYTES = NP * PIXB;
  static constexpr int KEY_SHIFT = (NC == 16) ? 0 : (NC == 8 ? 1 : 2);
  static constexpr int HITERS = (NP * NC + 255) / 256;
  // epilogue: the 128*WM x CT output tile is transposed through LDS ([pixel][CT] bf16, padded pitch) so that
  // global stores / residual loads are 16-byte pieces with consecutive lanes on consecutive addresses
  static constexpr int MPX = 32 * PXF * WM;
  static constexpr int EPITCH = CT * 2 + 16;
  static constexpr int EPI_BYTES = MPX * EPITCH;
  static constexpr int ENC = CT / 8;                      // 16-byte pieces per pixel of the tile
  static constexpr int EITERS = MPX * ENC / 256;
  static constexpr int STAT_OFF = HALO_BYTES > EPI_BYTES ? HALO_BYTES : EPI_BYTES;   // 256 floats of group sums
  // residual tile (or the GroupNorm input of the fused backward): its own LDS region, filled by LDS-DMA
  // (global_load_lds_dwordx4, no registers) at kernel start so that it shares the halo loads' round trip instead of
  // costing a second one in the epilogue.  Lane-linear image [pixel][CT/8 pieces], swizzled on the SOURCE side.
  static constexpr int RT_OFF = STAT_OFF + 1024 + 256;   // accumulators (1 KiB) + float table of the input statistics
  static constexpr int RT_BYTES = MPX * CT * 2;
  // bias[CT] (fp32), copied from global once at kernel start: a per-lane global load in the epilogue is an L2 round trip that
  // hipcc waits for with vmcnt(0) right at its use -- 8-16 SERIALISED round trips per tile in the round-2 kernels
  static constexpr int BIAS_OFF = RT_OFF + RT_BYTES;
  // gamma[CT] | beta[CT] of the fused GroupNorm-backward epilogue (same reason: 16-36 serialised loads per tile).  Lives in
  // the tail of the halo region that the output tile does not cover when that tail is large enough (free once the main loop
  // is over; keeps <3,64,64,2> at 40,960 B = four workgroups per CU), in its own 8*CT bytes otherwise.
  static constexpr bool GT_IN_TAIL = HALO_BYTES - EPI_BYTES >= 8 * CT;
  static constexpr int GT_OFF = GT_IN_TAIL ? EPI_BYTES : BIAS_OFF + 4 * CT;
  static constexpr int LDS_BYTES = BIAS_OFF + 4 * CT + (GT_IN_TAIL ? 0 : 8 * CT);
};


// conv_ws.hip: weight-stationary persistent kernel, Cin = Cout = 128, 3x3, stride-1 gathers.  Returns 0 when it took the
// launch, 1 when the launch is not one it covers (the caller falls back to the v2 kernel), > 1 on refusal codes of launch2.
int launch_conv_ws128(const ConvArgs& a, hipStream_t st);

}  // namespace pti_conv
