// Implicit-GEMM 3x3 / 1x1 convolution for gfx950 on v_mfma_f32_32x32x16_bf16.
//
// Replaces nn.Conv2d inside MONAI's AEKLResBlock / AEKLDownsample / Upsample / SABlock linears
// (reference src/pti_ldm_vae/models/autoencoder.py:67-79 builds them),
// with the GroupNorm-affine(+SiLU) that precedes each conv folded into the LDS loader, the
// nearest-2x up-sample / asymmetric-pad stride-2 gather folded into the loader's addressing, and
// bias + residual add + next-GroupNorm statistics folded into the epilogue.
//
// Layout: activations NHWC bf16.  One workgroup (256 threads = 4 waves) owns an 8x16 output-pixel
// tile (M = 128 GEMM rows) x COUT_TILE output channels:
//   * the input halo tile ((8-1)*S+K) x ((16-1)*S+K) pixels x CK channels is staged ONCE into LDS
//     (16-byte pieces, XOR-swizzled by the halo column so ds_read_b128 is conflict free) and
//     re-used by all K*K taps;
//   * weights arrive pre-packed in MFMA-fragment order (pti_conv_pack_weights), streamed through a
//     2-deep LDS ring in steps of WBLK k-blocks (16 input channels x COUT_TILE), prefetched into
//     registers one step ahead;
//   * MFMA orientation is D[cout][pixel] (A = weights, B = pixels) so each lane ends up with 4
//     consecutive output channels of one pixel per accumulator quad -> 8-byte NHWC stores.
#include "conv_common.h"

namespace {
using namespace pti_conv;

constexpr int TH = 8, TW = 16;  // output tile (pixels)

__host__ __device__ constexpr int pick_cout_tile(int cout) { return cout % 128 == 0 ? 128 : (cout % 64 == 0 ? 64 : 32); }
__host__ __device__ constexpr int pick_ck(int cin, int stride2) {
  int ck = cin % 128 == 0 ? 128 : (cin % 64 == 0 ? 64 : 32);
  return (stride2 && ck > 64) ? 64 : ck;
}
// largest divisor of kbc with wblk*nt <= 16 (one weight step <= 16 KiB)
__host__ __device__ constexpr int pick_wblk(int kbc, int nt) {
  int best = 1;
  for (int w = 1; w <= kbc; ++w)
    if (kbc % w == 0 && w * nt <= 16) best = w;
  return best;
}

template <int KS, int S, int CK, int COUT_TILE>
struct Cfg {
  static constexpr int HH = (TH - 1) * S + KS, HW = (TW - 1) * S + KS;
  static constexpr int NP = HH * HW;        // halo pixels
  static constexpr int NC = CK / 8;         // 16-byte pieces per pixel
  static constexpr int PIXB = CK * 2;       // bytes per halo pixel
  static constexpr int NT = COUT_TILE / 32; // 32-wide cout fragments per tile
  static constexpr int KPC = CK / 16;       // k-blocks per tap
  static constexpr int KBC = KS * KS * KPC; // k-blocks per cin chunk
  static constexpr int WBLK = pick_wblk(KBC, NT);
  static constexpr int NSTEP = KBC / WBLK;
  static constexpr int STEPB = WBLK * NT * 1024;  // bytes per weight step
  static constexpr int WM = (NT >= 2) ? 2 : 4, WN = 4 / WM;
  static constexpr int PXF = 4 / WM;   // pixel fragments (32 px) per wave
  static constexpr int CF = NT / WN;   // cout fragments per wave
  static constexpr int HALO_BYTES = NP * PIXB;
  static constexpr int LDS_BYTES = HALO_BYTES + 2 * STEPB;
  static constexpr int KEY_SHIFT = (NC == 16) ? 0 : (NC == 8 ? 1 : 2);
  static constexpr int WPIECES = (STEPB / 16 + 255) / 256;  // 16-byte pieces per thread per step
  static constexpr int HITERS = (NP * NC + 255) / 256;
};

template <int KS, int S, int CK, int COUT_TILE, bool OPH>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(ConvArgs a) {
  if (a.out_f16) fp16_saturate_on();   // wave-uniform (kernel argument)
  using C = Cfg<KS, S, CK, COUT_TILE>;
  __shared__ __attribute__((aligned(16))) unsigned char smem[C::LDS_BYTES];
  unsigned char* halo = smem;
  unsigned char* wbuf = smem + C::HALO_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hsel = lane >> 5;  // k-half selector of the MFMA operand maps
  int t = blockIdx.x;
  const int tile_x = t % a.tiles_x;
  t /= a.tiles_x;
  const int tile_y = t % a.tiles_y;
  const int n = t / a.tiles_y;
  const int oy0 = tile_y * TH, ox0 = tile_x * TW;
  const int ct = blockIdx.y;
  const int nchunks = a.Cin / CK;
  const int wm = wave / C::WN, wn = wave % C::WN;

  // virtual-input origin of the halo tile
  const int pad_lo = (a.mode == PTI_CONV_S2PAD) ? 0 : (a.mode == PTI_CONV_ZINS ? 2 : (KS - 1) / 2);
  const int vy0 = oy0 * S - pad_lo, vx0 = ox0 * S - pad_lo;
  const bool twox = (a.mode == PTI_CONV_UP2) || (a.mode == PTI_CONV_ZINS);
  const int VH = twox ? 2 * a.H : a.H, VW = twox ? 2 * a.W : a.W;

  // ---- per-lane LDS read addresses of the pixel (B) fragments ----
  int pbase[C::PXF][KS];
  int tkey[KS];
  {
    const int j = lane & 31;
    const int tx = j & 15;
#pragma unroll
    for (int kw = 0; kw < KS; ++kw) {
      const int hx = tx * S + kw;
      tkey[kw] = (hsel ^ ((hx >> C::KEY_SHIFT) & (C::NC - 1))) << 4;
#pragma unroll
      for (int i = 0; i < C::PXF; ++i) {
        const int ty = 2 * (wm * C::PXF + i) + (j >> 4);
        pbase[i][kw] = ((ty * S) * C::HW + hx) * C::PIXB;
      }
    }
  }

  f32x16 acc[C::PXF][C::CF];
#pragma unroll
  for (int i = 0; i < C::PXF; ++i)
#pragma unroll
    for (int c = 0; c < C::CF; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][c][r] = 0.f;

  // ---- weight ring: step g of this (cout tile) lives at wsrc + g*STEPB ----
  const unsigned char* wsrc = a.w + (size_t)ct * nchunks * C::NSTEP * C::STEPB;
  const int total_steps = nchunks * C::NSTEP;
  u32x4 wreg[C::WPIECES];
  auto wload = [&](int g) {
#pragma unroll
    for (int k = 0; k < C::WPIECES; ++k) {
      const int q = tid + k * 256;
      if (q * 16 < C::STEPB) wreg[k] = *(const u32x4*)(wsrc + (size_t)g * C::STEPB + q * 16);
    }
  };
  auto wstore = [&](int buf) {
#pragma unroll
    for (int k = 0; k < C::WPIECES; ++k) {
      const int q = tid + k * 256;
      if (q * 16 < C::STEPB) *(u32x4*)(wbuf + buf * C::STEPB + q * 16) = wreg[k];
    }
  };
  wload(0);
  wstore(0);

  // ---- halo loader geometry (constant per thread) ----
  const int lc = tid % C::NC;       // 16-byte piece (8 channels) within a pixel
  const int lp0 = tid / C::NC;      // first halo pixel of this thread
  constexpr int PSTEP = 256 / C::NC;
  const int cpg = a.Cin / (a.groups > 0 ? a.groups : 1);

  for (int chunk = 0; chunk < nchunks; ++chunk) {
    // ================= stage the halo tile of this cin chunk =================
    float sc[8], sh[8];
    if (a.prologue != PTI_PRO_NONE) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ch = chunk * CK + lc * 8 + j;
        const int g = ch / cpg;
        const float sum = stat_f(a.in_stats, (n * a.groups + g) * 2), sq = stat_f(a.in_stats, (n * a.groups + g) * 2 + 1);
        const float mean = sum * a.inv_cnt;
        const float var = fmaxf(sq * a.inv_cnt - mean * mean, 0.f);
        const float rstd = rsqrtf(var + a.eps);
        sc[j] = rstd * a.gamma[ch];
        sh[j] = a.beta[ch] - mean * sc[j];
      }
    }
    u32x4 raw[C::HITERS];
    bool ok[C::HITERS];
#pragma unroll
    for (int it = 0; it < C::HITERS; ++it) {
      const int p = lp0 + it * PSTEP;
      const int hy = p / C::HW, hx = p - hy * C::HW;
      const int vy = vy0 + hy, vx = vx0 + hx;
      bool v = (p < C::NP) && vy >= 0 && vy < VH && vx >= 0 && vx < VW;
      int iy = vy, ix = vx;
      if (twox) {
        if (a.mode == PTI_CONV_ZINS) v = v && !((vy | vx) & 1);
        iy = vy >> 1;
        ix = vx >> 1;
      }
      ok[it] = v;
      raw[it] = u32x4{0u, 0u, 0u, 0u};
      if (v) raw[it] = *(const u32x4*)(a.x + ((size_t)(n * a.H + iy) * a.W + ix) * a.Cin + chunk * CK + lc * 8);
    }
#pragma unroll
    for (int it = 0; it < C::HITERS; ++it) {
      const int p = lp0 + it * PSTEP;
      if (p < C::NP) {
        const int hy = p / C::HW, hx = p - hy * C::HW;
        u32x4 r = raw[it];
        if (a.prologue != PTI_PRO_NONE && ok[it]) {
          float f[8];
          unpack8f(r, f, a.in_f16);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float v = f[j] * sc[j] + sh[j];
            if (a.prologue == PTI_PRO_GN_SILU) v = silu_f(v);
            f[j] = v;
          }
          r = OPH ? pack8f(f, true) : pack8(f);
        } else if (a.in_f16 && !OPH) {   // no prologue: the MFMA operand is bf16, convert the fp16 piece
          float f[8];
          unpack8f(r, f, true);
          r = pack8(f);
        }
        const int key = (hx >> C::KEY_SHIFT) & (C::NC - 1);
        *(u32x4*)(halo + p * C::PIXB + ((lc ^ key) << 4)) = r;
      }
    }
    __syncthreads();

    // ================= MFMA main loop over the weight steps of this chunk =================
#pragma unroll
    for (int s = 0; s < C::NSTEP; ++s) {
      const int g = chunk * C::NSTEP + s;
      const bool more = (g + 1 < total_steps);
      if (more) wload(g + 1);
      const unsigned char* wcur = wbuf + (g & 1) * C::STEPB;
#pragma unroll
      for (int kbl = 0; kbl < C::WBLK; ++kbl) {
        constexpr int dummy = 0;
        (void)dummy;
        const int kb = s * C::WBLK + kbl;
        const int tap = kb / C::KPC, kc = kb % C::KPC;
        const int kh = tap / KS, kw = tap % KS;
        bf16x8 bfrag[C::PXF], afrag[C::CF];
#pragma unroll
        for (int i = 0; i < C::PXF; ++i)
          bfrag[i] = *(const bf16x8*)(halo + pbase[i][kw] + kh * C::HW * C::PIXB + ((kc * 32) ^ tkey[kw]));
#pragma unroll
        for (int c = 0; c < C::CF; ++c)
          afrag[c] = *(const bf16x8*)(wcur + ((kbl * C::NT + wn * C::CF + c) * 64 + lane) * 16);
#pragma unroll
        for (int i = 0; i < C::PXF; ++i)
#pragma unroll
          for (int c = 0; c < C::CF; ++c)
            acc[i][c] = mfma32<OPH>(afrag[c], bfrag[i], acc[i][c]);
      }
      if (more) wstore((g + 1) & 1);
      __syncthreads();
    }
  }

  // ================= epilogue: bias, residual, store, optional GN statistics =================
  const int j = lane & 31;
  stat_t* sstat = reinterpret_cast<stat_t*>(smem);  // [out_groups][2] fixed-point, reuse of the (now idle) halo
  const bool do_stats = a.out_stats != nullptr;
  const int ocpg = do_stats ? a.Cout / a.out_groups : 1;
  if (do_stats) {
    if (tid < 2 * a.out_groups) sstat[tid] = 0;
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < C::PXF; ++i) {
    const int ty = 2 * (wm * C::PXF + i) + (j >> 4), tx = j & 15;
    const int oy = oy0 + ty, ox = ox0 + tx;
    const bool inb = (oy < a.Ho) && (ox < a.Wo);
    const size_t pix = ((size_t)(n * a.Ho + oy) * a.Wo + ox) * a.Cout;
#pragma unroll
    for (int c = 0; c < C::CF; ++c) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co = ct * COUT_TILE + (wn * C::CF + c) * 32 + 8 * q + 4 * hsel;
        float v0 = acc[i][c][4 * q + 0], v1 = acc[i][c][4 * q + 1], v2 = acc[i][c][4 * q + 2],
              v3 = acc[i][c][4 * q + 3];
        if (a.bias) {
          const f32x4 b = *(const f32x4*)(a.bias + co);
          v0 += b[0]; v1 += b[1]; v2 += b[2]; v3 += b[3];
        }
        u32x2 packed = u32x2{0u, 0u};
        if (inb) {
          if (a.res) {
            const u32x2 rr = *(const u32x2*)(a.res + pix + co);
            float e0, e1, e2, e3;
            unpack2f(rr[0], a.res_f16, e0, e1);
            unpack2f(rr[1], a.res_f16, e2, e3);
            v0 += e0; v1 += e1; v2 += e2; v3 += e3;
          }
          packed = pack4f(v0, v1, v2, v3, a.out_f16);
          *(u32x2*)(a.y + pix + co) = packed;
        }
        if (do_stats) {
          // statistics of the values as stored (rounded to 16 bits), like a later read pass would see
          float r0, r1, r2, r3;
          unpack2f(packed[0], a.out_f16, r0, r1);
          unpack2f(packed[1], a.out_f16, r2, r3);
          if (!inb) r0 = r1 = r2 = r3 = 0.f;
          if (ocpg >= 4) {
            float s1 = (r0 + r1) + (r2 + r3), s2 = (r0 * r0 + r1 * r1) + (r2 * r2 + r3 * r3);
            // lanes 0..31 (and 32..63) of this quad hold 32 different pixels of the same 4 channels
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) {
              s1 += __shfl_xor(s1, o, 64);
              s2 += __shfl_xor(s2, o, 64);
            }
            if (j == 0) {
              const int g = co / ocpg;
              stat_add(&sstat[2 * g], s1);
              stat_add(&sstat[2 * g + 1], s2);
            }
          } else {  // ocpg == 2: two groups inside the quad
            float a1 = r0 + r1, a2 = r0 * r0 + r1 * r1, b1 = r2 + r3, b2 = r2 * r2 + r3 * r3;
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) {
              a1 += __shfl_xor(a1, o, 64);
              a2 += __shfl_xor(a2, o, 64);
              b1 += __shfl_xor(b1, o, 64);
              b2 += __shfl_xor(b2, o, 64);
            }
            if (j == 0) {
              const int g = co / 2;
              stat_add(&sstat[2 * g], a1);
              stat_add(&sstat[2 * g + 1], a2);
              stat_add(&sstat[2 * g + 2], b1);
              stat_add(&sstat[2 * g + 3], b2);
            }
          }
        }
      }
    }
  }
  if (do_stats) {
    __syncthreads();
    const int g0 = (ct * COUT_TILE) / ocpg, ng = COUT_TILE / ocpg;
    if (tid < 2 * ng)
      atomicAdd((unsigned long long*)&a.out_stats[(n * a.out_groups + g0) * 2 + tid], (unsigned long long)sstat[2 * g0 + tid]);
  }
}

// =============================================================================================
// v2 kernel (stride-1 gathers: PTI_CONV_S1 / UP2 / ZINS, k in {1,3}).
// Measured on v1: LDS was the bottleneck (weights written to + read from LDS every step, one barrier
// per step).  v2 keeps ONLY the halo tile in LDS.  Each wave owns 128 pixels (4 MFMA pixel fragments)
// x 32 output channels, and streams its weight fragments straight from global/L2 into an 8-deep
// register ring (the packed layout makes every fragment one coalesced 1-KiB wave load), so the main
// loop has no LDS writes and no barriers; 4 MFMAs are issued per 4 ds_read_b128 + 1 global load.
// Workgroup = 4 waves = (4/WN) pixel groups x WN cout fragments, i.e. 32*PXF*WM pixels x CT channels.
// PXF (MFMA pixel fragments per wave) is 4 for the 128-channel tiles (MFMA-bound, 2 workgroups/CU) and 2 for
// the 32/64-channel tiles: those layers are HBM-latency-bound, and halving the accumulators + a 3-deep weight
// ring fits 4 workgroups/CU (128 VGPRs) -- measured 32->32@256^2 100.6 -> 81.4 us, 64->32 159 -> 133 us.
// (tried: PXF = 1 -> more weight traffic per MFMA and 1.4-1.7x halo amplification, 1.5x slower.)
// 128-channel tile (128->128 @64^2, batch 32: ~46-50 us, ~800 TF/s), what was measured:
//  * no global halo loads and no stores: 35 us; also no weight loads / LDS reads in the loop: 30 us (MFMA skeleton);
//    staging + epilogue alone (no MFMA loop): 21 us.  The memory phases are NOT hidden behind the MFMA loop: a
//    launch is only 2 rounds of 2 workgroups/CU and every workgroup is in the same phase at the same time.
//  * 1-row x 32-pixel fragments shared across the 3 kernel rows (-40 % LDS reads): no gain (LDS ~21 % busy).
//  * PXF = 2 at 3-4 workgroups/CU, weight ring 3/6/9 deep: all within +-3 %.  A start delay for every other
//    workgroup: slower.  64-channel chunks with the next chunk's halo prefetched into registers under the current
//    chunk's MFMAs (246-253 VGPRs): no gain.  Next: persistent workgroups with cross-tile halo prefetch.
//  * (round 2) the 32^2 maps at batch 32 are only 256 workgroups of 128 pixels (one per CU, 21-23 us per launch):
//    re-tiling those launches with PXF = 2 (512 workgroups of 64 pixels, no spills, 3 resident per CU) left the
//    training step unchanged (12.04 vs 12.04 ms, same box), so it is not in.
// =============================================================================================
// FM = storage-format mode: 0 = the three 16-bit formats are run-time flags; 1 = input, residual and output are all
// fp16 (every forward conv of the default engine), 2 = all bf16 (plain data gradients), 3 = bf16 in / fp16 residual /
// bf16 out (data gradient fused with the GroupNorm backward: the "residual" is the fp16 GN input).  Known formats
// drop the convert-both-ways-and-select per element (the narrow-layer convs are VALU-bound).
// PRO = the prologue when known at compile time (PTI_PRO_NONE / PTI_PRO_GN_SILU), -1 = run-time.  Modes 2 and 3 are
// data gradients (no prologue); mode 3 IS the fused GroupNorm-backward epilogue, modes 1 and 2 never are: the
// unused prologue / epilogue code and its registers (16 scale/shift VGPRs at the 128-VGPR cap) disappear.
template <int KS, int CK, int CT, int PXF, bool SAVE, int FM, int PRO>
__global__ __launch_bounds__(256, (PXF == 2 ? 4 : 2)) void conv_mfma2_kernel(ConvArgs a) {
  using C = Cfg2<KS, CK, CT, PXF>;
  const int prologue = PRO >= 0 ? PRO : a.prologue;
  const bool gn_on = FM == 3 ? true : (FM == 0 ? a.gn_mode != 0 : false);
  const bool in_f16 = FM == 0 ? (bool)a.in_f16 : (FM == 1);
  const bool res_f16 = FM == 0 ? (bool)a.res_f16 : (FM == 1 || FM == 3);
  const bool out_f16 = FM == 0 ? (bool)a.out_f16 : (FM == 1);
  // FM == 1 (every operand tensor fp16: the forward convs of the default engine) multiplies fp16 operands
  // (v_mfma_f32_32x32x16_f16, fp16-packed weights): same rate as bf16, 8x finer operand rounding, and the fp16
  // activations need no conversion on their way into LDS.  Everything else (gradients) stays bf16.
  constexpr bool OPH = (FM == 1);
  if (out_f16) fp16_saturate_on();   // wave-uniform
  __shared__ __attribute__((aligned(16))) unsigned char smem[C::LDS_BYTES];
  unsigned char* halo = smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hsel = lane >> 5;
  int t = blockIdx.x;
  const int tile_x = t % a.tiles_x;
  t /= a.tiles_x;
  const int tile_y = t % a.tiles_y;
  const int n = t / a.tiles_y;
  const int oy0 = tile_y * C::TH2, ox0 = tile_x * C::TW2;
  const int ct = blockIdx.y;
  const int nchunks = a.Cin / CK;
  const int wm = wave / C::WN, wn = wave % C::WN;

  const int pad_lo = (a.mode == PTI_CONV_ZINS) ? 2 : (KS - 1) / 2;
  const int vy0 = oy0 - pad_lo, vx0 = ox0 - pad_lo;
  const bool twox = (a.mode == PTI_CONV_UP2) || (a.mode == PTI_CONV_ZINS);
  const int VH = twox ? 2 * a.H : a.H, VW = twox ? 2 * a.W : a.W;

  // per-lane LDS addresses of the pixel (B) fragments: fragment i = tile rows 8*wm + 2i, +1
  int pbase[KS], tkey[KS];
  {
    const int j = lane & 31, tx = j & 15, row0 = 2 * PXF * wm + (j >> 4);
#pragma unroll
    for (int kw = 0; kw < KS; ++kw) {
      const int hx = tx + kw;
      tkey[kw] = (hsel ^ ((hx >> C::KEY_SHIFT) & (C::NC - 1))) << 4;
      pbase[kw] = (row0 * C::HW + hx) * C::PIXB;
    }
  }
  f32x16 acc[PXF];
#pragma unroll
  for (int i = 0; i < PXF; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  const int lc = tid % C::NC, lp0 = tid / C::NC;
  constexpr int PSTEP = 256 / C::NC;
  const int cpg = a.Cin / (a.groups > 0 ? a.groups : 1);
  reinterpret_cast<float*>(smem + C::STAT_OFF)[tid] = 0.f;   // visible after the first barrier below
  // this sample's GroupNorm sums (of the prologue's input, or of the fused backward's GN input) as floats in LDS:
  // 2*G threads convert the fixed-point values once instead of every thread converting the ones its channels need
  // (256 B behind the 1-KiB accumulator area, so groups <= 32)
  float* sfl = reinterpret_cast<float*>(smem + C::STAT_OFF + 1024);
  if (prologue == PTI_PRO_GN || prologue == PTI_PRO_GN_SILU) {
    if (tid < 2 * a.groups) sfl[tid] = stat_f(a.in_stats, n * a.groups * 2 + tid);
  } else if (gn_on) {
    if (tid < 2 * a.g_groups) sfl[tid] = stat_f(a.g_stats, n * a.g_groups * 2 + tid);
  }
  // epilogue parameters: bias -> LDS now (the first barrier below publishes it); gamma / beta of the fused GroupNorm
  // backward -> two registers of the first CT threads now, LDS after the main loop (their slot may alias the halo tile)
  float gq = 0.f, bq = 0.f;
  if (gn_on) {
    if (tid < CT) {
      gq = a.g_gamma[ct * CT + tid];
      bq = a.g_beta[ct * CT + tid];
    }
  } else if (a.bias) {
    if (tid < CT) reinterpret_cast<float*>(smem + C::BIAS_OFF)[tid] = a.bias[ct * CT + tid];
  }
  if (a.res) {
    // piece (p, c) of the residual tile -> LDS slot p*ENC + c, holding channel piece c ^ ((p >> 2) & (ENC-1)) (the
    // epilogue reads apply the same XOR: 8-byte reads of one piece column at a 64..256-byte pixel pitch would
    // otherwise hit the same banks).  Out-of-image pixels are skipped (their slots are never used for output).
#pragma unroll
    for (int it = 0; it < C::EITERS; ++it) {
      const int slot = it * 256 + tid;
      const int p = slot / C::ENC, c = slot % C::ENC;
      const int oy = oy0 + p / C::TW2, ox = ox0 + p % C::TW2;
      if (oy < a.Ho && ox < a.Wo) {
        const bf16* src = a.res + ((size_t)(n * a.Ho + oy) * a.Wo + ox) * a.Cout + ct * CT + ((c ^ ((p >> 2) & (C::ENC - 1))) << 3);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(smem + C::RT_OFF + (it * 256 + wave * 64) * 16),
                                         16, 0, 0);
      }
    }
  }

  // (tried: fetching the residual tile to registers under the last chunk's MFMA loop -> spills at the
  //  128-VGPR cap of the 4-workgroup/CU shapes, 1.9x slower)
  const int epc = tid % C::ENC, epp0 = tid / C::ENC;
  constexpr int EPSTEP = 256 / C::ENC;

  for (int chunk = 0; chunk < nchunks; ++chunk) {
    // weight fragments of this (cout tile, cin chunk): [kb][nt][lane][8]; start the ring first so the
    // loads fly while the halo tile is staged
    const unsigned char* wlane =
        a.w + ((size_t)(ct * nchunks + chunk) * C::KBC * C::NT + wn) * 1024 + lane * 16;
    // two alternating register sets of R weight fragments: set A is consumed while set B (the next group of
    // R k-blocks) is in flight, so every global load has a whole group (R x 4 MFMAs) to land
    bf16x8 wa[C::R], wb[C::R];
    auto wload = [&](bf16x8 (&dst)[C::R], int g) {
#pragma unroll
      for (int u = 0; u < C::R; ++u) {
        int kb = g * C::R + u;
        kb = kb < C::KBC ? kb : C::KBC - 1;
        dst[u] = *(const bf16x8*)(wlane + (size_t)kb * C::NT * 1024);
      }
    };
    // (side-output variant at the 128-VGPR cap, and the two-input prologue with its second set of staging registers: after staging)
    constexpr bool WLATE = (SAVE && PXF == 2) || PRO == PRO_GNB;
    if constexpr (!WLATE) wload(wa, 0);

    // Halo loads through a buffer descriptor with 32-bit byte offsets: an out-of-image (or zero-inserted, or
    // past-the-tile) piece gets an offset beyond the descriptor's range and the hardware returns zeros -- no
    // per-piece branch, no 64-bit address arithmetic (the generic-address form spent ~50 VALU instructions per piece,
    // 18 of them quarter-rate v_mad_u64_u32, on kernels that are VALU-bound: ~1/5 of a narrow-layer tile's VALU work).
    // The host guarantees x has fewer than 2^31 bytes.
    u32x4 raw[C::HITERS];
    u32x4 raw2[PRO == PRO_GNB ? C::HITERS : 1];     // PRO_GNB: the same pieces of the second input (the GroupNorm input)
    bool ok[C::HITERS];
    {
      const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<bf16*>(a.x), 0, (unsigned)a.N * (unsigned)a.H * (unsigned)a.W * (unsigned)a.Cin * 2u, 0x00020000);
      const __amdgpu_buffer_rsrc_t x2rs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<bf16*>(PRO == PRO_GNB ? a.x2 : a.x), 0, (unsigned)a.N * (unsigned)a.H * (unsigned)a.W * (unsigned)a.Cin * 2u, 0x00020000);
      const unsigned base = ((unsigned)n * a.H * a.W * a.Cin + chunk * CK + lc * 8) * 2u;   // sample + channel piece
      const unsigned rowb = (unsigned)a.W * a.Cin * 2u, pixb = (unsigned)a.Cin * 2u;
#pragma unroll
      for (int it = 0; it < C::HITERS; ++it) {
        const int p = lp0 + it * PSTEP;
        const int hy = p / C::HW, hx = p - hy * C::HW;
        const int vy = vy0 + hy, vx = vx0 + hx;
        bool v = (p < C::NP) && (unsigned)vy < (unsigned)VH && (unsigned)vx < (unsigned)VW;
        int iy = vy, ix = vx;
        if (twox) {
          if (a.mode == PTI_CONV_ZINS) v = v && !((vy | vx) & 1);
          iy = vy >> 1;
          ix = vx >> 1;
        }
        ok[it] = v;
        const unsigned off = v ? base + (unsigned)iy * rowb + (unsigned)ix * pixb : 0x80000000u;
        raw[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0));
        if constexpr (PRO == PRO_GNB) raw2[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(x2rs, off, 0, 0));
      }
    }
    // GroupNorm scale / shift of this thread's 8 channels: fetched AFTER the halo loads were issued, so that the
    // statistics / gamma / beta round trips (L2) overlap the halo's HBM round trip instead of preceding it
    if (chunk == 0) __syncthreads();   // the float statistics table (sfl) is complete
    float sc[8], sh[8];
    [[maybe_unused]] float gnb_b = 0.f, gnb_d = 0.f;
    if constexpr (PRO == PRO_GNB) {
      // staged value = rstd*(gamma*g - c1 - xhat*c2) = sc[j]*g + gnb_b*h + gnb_d with xhat = (h - mean)*rstd and, per
      // (sample, group), c1 = mean_c(gamma*sum g), c2 = mean_c(gamma*sum g*xhat) over the group's channels x pixels.
      // One 8-channel piece lies inside one group (channels per group >= 8); a wider group is walked.
      const int ch0 = chunk * CK + lc * 8, g = ch0 / cpg;
      const float mean = stat_f(a.in_stats, (n * a.groups + g) * 2) * a.inv_cnt;
      const float rstd = rsqrtf(fmaxf(stat_f(a.in_stats, (n * a.groups + g) * 2 + 1) * a.inv_cnt - mean * mean, 0.f) + a.eps);
      float t1 = 0.f, t2 = 0.f;
      for (int cc = g * cpg; cc < (g + 1) * cpg; ++cc) {
        const f32x2 sv = *(const f32x2*)(a.p_sums + ((size_t)n * a.Cin + cc) * 2);
        const float gm = a.gamma[cc];
        t1 += gm * sv[0];
        t2 += gm * sv[1];
      }
      const float c1 = t1 * a.inv_cnt, c2 = t2 * a.inv_cnt;
#pragma unroll
      for (int j = 0; j < 8; ++j) sc[j] = rstd * a.gamma[ch0 + j];
      gnb_b = -rstd * rstd * c2;
      gnb_d = -rstd * c1 - gnb_b * mean;
    } else if (prologue != PTI_PRO_NONE) {
      // per GROUP on a straight-line path per group size (shift for the group index, one v_rsq per group): the
      // per-channel form -- integer division by the run-time channels-per-group and a guarded rsqrtf per channel --
      // cost ~25 VALU instructions per channel on kernels that are VALU-bound
      float mu_[8], rs_[8];
      gn_params<8>(sfl, a.gamma, a.beta, chunk * CK + lc * 8, cpg, a.inv_cnt, a.eps, sc, sh, mu_, rs_);
    }
    if (chunk > 0) __syncthreads();  // every wave is done reading the previous chunk's halo
#pragma unroll
    for (int it = 0; it < C::HITERS; ++it) {
      const int p = lp0 + it * PSTEP;
      if (p < C::NP) {
        const int hy = p / C::HW, hx = p - hy * C::HW;
        u32x4 r = raw[it];
        if constexpr (PRO == PRO_GNB) {
          if (ok[it]) {
            float f[8], h[8];
            unpack8(r, f);
            unpack8f(raw2[it], h, a.x2_f16 != 0);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = fmaf(sc[j], f[j], fmaf(gnb_b, h[j], gnb_d));
            r = pack8(f);
          }
        } else if (prologue != PTI_PRO_NONE && ok[it]) {
          float f[8];
          unpack8f(r, f, in_f16);
          if (PXF == 4 && prologue == PTI_PRO_GN_SILU) {
            // packed-fp32 form: -3..-5 % on the 2-workgroup/CU shapes; at the 128-VGPR cap of the others it spills
            // (32->32@256^2 +res+stats 152 -> 162 us), they take the scalar form below
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
              const f32x2 o = gn_silu2(f32x2{f[j], f[j + 1]}, f32x2{sc[j], sc[j + 1]}, f32x2{sh[j], sh[j + 1]});
              f[j] = o[0];
              f[j + 1] = o[1];
            }
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              float v = f[j] * sc[j] + sh[j];
              if (prologue == PTI_PRO_GN_SILU) v = silu_f(v);
              f[j] = v;
            }
          }
          r = OPH ? pack8f(f, true) : pack8(f);
        } else if (in_f16 && !OPH) {   // no prologue: the MFMA operand is bf16, convert the fp16 piece
          float f[8];
          unpack8f(r, f, true);
          r = pack8(f);
        }
        const int key = (hx >> C::KEY_SHIFT) & (C::NC - 1);
        *(u32x4*)(halo + p * C::PIXB + ((lc ^ key) << 4)) = r;
      }
    }
    __syncthreads();

    if constexpr (SAVE) {
      // side output act(GN(x)): copied out of the staged tile (interior pixels only; halo pixels belong to the
      // neighbouring tiles) rather than stored from the staging registers -- that kept addresses live across the
      // staging loop and spilled ~17 VGPRs at the 128-register cap (+300 MB of scratch traffic per launch)
      if (ct == 0) {
        constexpr int PL = (KS - 1) / 2;
        constexpr int SITERS = C::TH2 * C::TW2 * C::NC / 256;
        const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(
            a.act_out, 0, (unsigned)a.N * (unsigned)a.H * (unsigned)a.W * (unsigned)a.Cin * 2u, 0x00020000);
#pragma unroll 1
        for (int it = 0; it < SITERS; ++it) {
          const int idx = tid + it * 256;
          const int c8 = idx % C::NC, pi = idx / C::NC;
          const int hy = pi / C::TW2 + PL, hx = pi % C::TW2 + PL;
          const int vy = vy0 + hy, vx = vx0 + hx;
          if (vy < a.H && vx < a.W) {
            const int key = (hx >> C::KEY_SHIFT) & (C::NC - 1);
            u32x4 piece = *(const u32x4*)(halo + (hy * C::HW + hx) * C::PIXB + ((c8 ^ key) << 4));
            if constexpr (OPH) {   // the staged operand is fp16; the saved copy is the weight gradient's bf16 operand
              float f[8];
              unpack8f(piece, f, true);
              piece = pack8(f);
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(bu32x4, piece), srs,
                                                   ((unsigned)((n * a.H + vy) * a.W + vx) * a.Cin + chunk * CK + c8 * 8) * 2u, 0, 2 /* nt */);
          }
        }
      }
    }

    if constexpr (WLATE) wload(wa, 0);
    // main loop over groups of R k-blocks.  B (pixel) fragments are double-buffered in registers: the 4
    // ds_read_b128 of k-block u+1 are issued before the 4 MFMAs of k-block u.
    const int txl = lane & 15;
    auto baddr = [&](int kb) -> const unsigned char* {
      kb = kb < C::KBC ? kb : C::KBC - 1;
      const int tap = kb / C::KPC, kc = kb - tap * C::KPC;
      const int kh = (KS == 3) ? tap / 3 : 0, kw = tap - kh * KS;
      const int hx = txl + kw;
      const int tk = (hsel ^ ((hx >> C::KEY_SHIFT) & (C::NC - 1))) << 4;
      return halo + pbase[0] + (kh * C::HW + kw) * C::PIXB + ((kc * 32) ^ tk);
    };
    bf16x8 b0[PXF], b1[PXF];
    auto bread = [&](bf16x8 (&dst)[PXF], int kb) {
      const unsigned char* bp = baddr(kb);
#pragma unroll
      for (int i = 0; i < PXF; ++i) dst[i] = *(const bf16x8*)(bp + 2 * i * C::HW * C::PIXB);
    };
    auto group = [&](const bf16x8 (&w)[C::R], int g) {
      const int kb0 = g * C::R;
#pragma unroll
      for (int u = 0; u < C::R; ++u) {
        if ((u & 1) == 0) {
          bread(b1, kb0 + u + 1);
#pragma unroll
          for (int i = 0; i < PXF; ++i) acc[i] = mfma32<OPH>(w[u], b0[i], acc[i]);
        } else {
          bread(b0, kb0 + u + 1);
#pragma unroll
          for (int i = 0; i < PXF; ++i) acc[i] = mfma32<OPH>(w[u], b1[i], acc[i]);
        }
      }
      if (C::R & 1) {  // odd group length: the last read went to b1/b0 alternately; realign so b0 is current
#pragma unroll
        for (int i = 0; i < PXF; ++i) b0[i] = b1[i];
      }
    };
    constexpr int NG = C::KBC / C::R;
    bread(b0, 0);
    if constexpr (NG == 1) {
      group(wa, 0);
    } else {
#pragma clang loop unroll(disable)
      for (int g = 0; g < NG; g += 2) {
        wload(wb, g + 1);
        group(wa, g);
        wload(wa, g + 2);
        group(wb, g + 1);
      }
    }
  }

  // ---- epilogue: bias, residual, bf16 rounding, optional GroupNorm statistics, coalesced store ----
  const int j = lane & 31;
  unsigned char* etile = smem;
  const bool do_stats = (FM == 2 || FM == 3) ? false : a.out_stats != nullptr;   // data gradients feed no GroupNorm
  const int ocpg = do_stats ? a.Cout / a.out_groups : 1;
  __syncthreads();  // every wave is done with the halo tile
  float* gtab = reinterpret_cast<float*>(smem + C::GT_OFF);   // gamma[CT] | beta[CT]
  if (gn_on) {
    if (tid < CT) {
      gtab[tid] = gq;
      gtab[CT + tid] = bq;
    }
    __syncthreads();
  }
  // (A) the residual tile is already in LDS (rtile, LDS-DMA issued at kernel start; every barrier since drained it)
  const unsigned char* rtile = smem + C::RT_OFF;
  const int col0 = wn * 32 + 4 * hsel;   // channel within the CT tile
  float st1[4] = {0.f, 0.f, 0.f, 0.f}, st2[4] = {0.f, 0.f, 0.f, 0.f};   // per quad q (or first pair when ocpg==2)
  float su1[4] = {0.f, 0.f, 0.f, 0.f}, su2[4] = {0.f, 0.f, 0.f, 0.f};   // second pair of the quad when ocpg==2
  if (gn_on) {
    // (B') data gradient + GroupNorm backward reduction: etile holds the GN input gx
    const int gcpg = a.Cout / a.g_groups;
    float* gsm = reinterpret_cast<float*>(smem + C::STAT_OFF);   // [WM][CT][2]: one slot per (pixel group, channel, sum)
    float L[32];   // [q][r][{sum dy, sum dy*xhat}] of this lane
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int col = col0 + 8 * q, ch0 = ct * CT + col;
      float scv[4], shv[4], muv[4], rsv[4], l1[4] = {0.f, 0.f, 0.f, 0.f}, l2[4] = {0.f, 0.f, 0.f, 0.f};
      // gamma / beta from the LDS copy (indexed by the channel within this cout tile; the group index still needs ch0)
      gn_params<4>(sfl, gtab - ct * CT, gtab + CT - ct * CT, ch0, gcpg, a.g_inv_cnt, a.g_eps, scv, shv, muv, rsv);
      float nmr[4];     // -mean * rstd: xhat = x * rstd + nmr
#pragma unroll
      for (int r = 0; r < 4; ++r) nmr[r] = -muv[r] * rsv[r];
#pragma unroll
      for (int i = 0; i < PXF; ++i) {
        const int p = (2 * PXF * wm + 2 * i + (j >> 4)) * 16 + (j & 15);
        const bool inb = (oy0 + (p >> 4) < a.Ho) && (ox0 + (p & 15) < a.Wo);
        unsigned char* ep = etile + p * C::EPITCH + col * 2;
        const u32x2 rr = *(const u32x2*)(rtile + p * (CT * 2) + ((((col >> 3) ^ (p >> 2)) & (C::ENC - 1)) << 4) + (col & 7) * 2);
        float xv[4];
        unpack2f(rr[0], res_f16, xv[0], xv[1]);
        unpack2f(rr[1], res_f16, xv[2], xv[3]);
        float dv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[i][4 * q + r];
          if (a.gn_mode == 2) v *= dsilu_f(xv[r] * scv[r] + shv[r]);
          dv[r] = v;
        }
        const u32x2 packed = pack4(dv[0], dv[1], dv[2], dv[3]);
        *(u32x2*)ep = packed;
        if (inb) {
          // sums of the fp32 values (not of their bf16 roundings: one unpack per value less on a VALU-bound epilogue; the
          // difference is the rounding error of a sum of N terms, ~2^-9 / sqrt(N) relative), xhat as ONE fma per value
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            l1[r] += dv[r];
            l2[r] = fmaf(dv[r], fmaf(xv[r], rsv[r], nmr[r]), l2[r]);
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        L[8 * q + 2 * r] = l1[r];
        L[8 * q + 2 * r + 1] = l2[r];
      }
    }
    fold32<32>(L, j);   // lane j now holds the half-wave total of value j = 8q + 2r + s, i.e. channel col0 + 8q + r, sum s
    // plain store into this wave's own slot (the waves of one channel range differ in wm): no atomics, so the
    // workgroup's partial -- summed over wm in a fixed order below -- is bitwise reproducible
    gsm[wm * 2 * CT + col0 * 2 + 16 * (j >> 3) + (j & 7)] = L[0];
  } else {
  // (B) registers -> (+bias, +residual) -> bf16 -> LDS tile; statistics accumulate in-lane over the 4 fragments
#pragma unroll
  for (int i = 0; i < PXF; ++i) {
    const int p = (2 * PXF * wm + 2 * i + (j >> 4)) * 16 + (j & 15);
    const bool inb = (oy0 + (p >> 4) < a.Ho) && (ox0 + (p & 15) < a.Wo);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int col = col0 + 8 * q;
      float v0 = acc[i][4 * q + 0], v1 = acc[i][4 * q + 1], v2 = acc[i][4 * q + 2], v3 = acc[i][4 * q + 3];
      if (a.bias) {
        const f32x4 b = *(const f32x4*)(smem + C::BIAS_OFF + col * 4);
        v0 += b[0]; v1 += b[1]; v2 += b[2]; v3 += b[3];
      }
      unsigned char* ep = etile + p * C::EPITCH + col * 2;
      if (a.res) {
        const u32x2 rr = *(const u32x2*)(rtile + p * (CT * 2) + ((((col >> 3) ^ (p >> 2)) & (C::ENC - 1)) << 4) + (col & 7) * 2);
        float e0, e1, e2, e3;
        unpack2f(rr[0], res_f16, e0, e1);
        unpack2f(rr[1], res_f16, e2, e3);
        v0 += e0; v1 += e1; v2 += e2; v3 += e3;
      }
      if constexpr (FM == 1 && PRO == PTI_PRO_NONE) {   // the only instantiation that carries the fused ReLU
        if (a.relu_out) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
      }
      const u32x2 packed = pack4f(v0, v1, v2, v3, out_f16);
      *(u32x2*)ep = packed;
      if (do_stats && inb) {
        float r0, r1, r2, r3;
        unpack2f(packed[0], out_f16, r0, r1);
        unpack2f(packed[1], out_f16, r2, r3);
        if (ocpg >= 4) {
          st1[q] += (r0 + r1) + (r2 + r3);
          st2[q] += (r0 * r0 + r1 * r1) + (r2 * r2 + r3 * r3);
        } else {
          st1[q] += r0 + r1; st2[q] += r0 * r0 + r1 * r1;
          su1[q] += r2 + r3; su2[q] += r2 * r2 + r3 * r3;
        }
      }
    }
  }
  }
  if (do_stats) {
    // fold the 32 pixel-lanes of each wave half (fold32), then LDS atomics: one lane per (channel quad, sum)
    stat_t* sstat = reinterpret_cast<stat_t*>(smem + C::STAT_OFF);   // fixed-point; zeroed before the main loop
    if (ocpg >= 4) {
      float v[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) { v[2 * q] = st1[q]; v[2 * q + 1] = st2[q]; }
      fold32<8>(v, j);
      if ((j & 3) == 0) {
        const int idx = j >> 2, q = idx >> 1;
        const int g = (ct * CT + col0 + 8 * q) / ocpg;
        stat_add(&sstat[2 * g + (idx & 1)], v[0]);
      }
    } else {   // two channels per group: every quad spans two groups
      float v[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) { v[4 * q] = st1[q]; v[4 * q + 1] = st2[q]; v[4 * q + 2] = su1[q]; v[4 * q + 3] = su2[q]; }
      fold32<16>(v, j);
      if ((j & 1) == 0) {
        const int idx = j >> 1, q = idx >> 2;
        const int g = (ct * CT + col0 + 8 * q) / 2;
        stat_add(&sstat[2 * g + (idx & 3)], v[0]);
      }
    }
  }
  __syncthreads();
  // (C) LDS tile -> global, 16 bytes per lane, consecutive lanes on consecutive addresses (buffer stores with 32-bit
  // offsets; non-temporal: the output is not re-read by this launch)
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
      a.y, 0, ((unsigned)a.N * (unsigned)a.Ho * (unsigned)a.Wo * (unsigned)a.Cout * 2u) >> (a.pool2 ? 2 : 0), 0x00020000);
  if ((FM == 0 || FM == 2) && a.pool2) {   // pooled output: plain data gradients only
    // data gradient of  conv(nearest-2x(x)): the gradient w.r.t. x is the 2x2 sum of the gradient w.r.t. the
    // up-sampled map -- summed here (fp32) from the LDS tile instead of writing the full-resolution map and
    // pooling it in a second pass.  Tile origins and sizes are even, so every 2x2 cell lies inside one tile.
    constexpr int PITERS = (C::EITERS + 3) / 4;
#pragma unroll
    for (int it = 0; it < PITERS; ++it) {
      const int idx = tid + it * 256;
      const int c8 = idx % C::ENC, pp = idx / C::ENC;
      const int py = pp / 8, px = pp % 8;
      const int oy = (oy0 >> 1) + py, ox = (ox0 >> 1) + px;
      if (pp < C::MPX / 4 && oy < (a.Ho >> 1) && ox < (a.Wo >> 1)) {
        const unsigned char* src = etile + ((2 * py) * 16 + 2 * px) * C::EPITCH + c8 * 16;
        float s_[8], f_[8];
        unpack8f(*(const u32x4*)src, s_, out_f16);
        unpack8f(*(const u32x4*)(src + C::EPITCH), f_, out_f16);
#pragma unroll
        for (int q = 0; q < 8; ++q) s_[q] += f_[q];
        unpack8f(*(const u32x4*)(src + 16 * C::EPITCH), f_, out_f16);
#pragma unroll
        for (int q = 0; q < 8; ++q) s_[q] += f_[q];
        unpack8f(*(const u32x4*)(src + 17 * C::EPITCH), f_, out_f16);
#pragma unroll
        for (int q = 0; q < 8; ++q) s_[q] += f_[q];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(bu32x4, pack8f(s_, out_f16)), yrs,
                                               ((unsigned)((n * (a.Ho >> 1) + oy) * (a.Wo >> 1) + ox) * a.Cout + ct * CT + c8 * 8) * 2u, 0, 2);
      }
    }
  } else
#pragma unroll
  for (int it = 0; it < C::EITERS; ++it) {
    const int p = epp0 + it * EPSTEP;
    const int oy = oy0 + (p >> 4), ox = ox0 + (p & 15);
    if (oy < a.Ho && ox < a.Wo)
      // non-temporal: the output is not re-read by this launch; keeping it out of L2's way measured -1.1 % per step
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(bu32x4, *(const u32x4*)(etile + p * C::EPITCH + epc * 16)), yrs,
                                             ((unsigned)((n * a.Ho + oy) * a.Wo + ox) * a.Cout + ct * CT + epc * 8) * 2u, 0, 2);
  }
  if (gn_on) {
    if (tid < 2 * CT) {
      const float* gsm = reinterpret_cast<const float*>(smem + C::STAT_OFF);
      float v = gsm[tid];
#pragma unroll
      for (int m = 1; m < C::WM; ++m) v += gsm[m * 2 * CT + tid];
      // partial of this pixel tile: [n][tile][channel][sum], one coalesced plain store per workgroup
      a.g_sums[(((size_t)n * a.g_T + tile_y * a.tiles_x + tile_x) * a.Cout + ct * CT) * 2 + tid] = v;
    }
  } else if (do_stats) {
    const stat_t* sstat = reinterpret_cast<const stat_t*>(smem + C::STAT_OFF);
    const int g0 = (ct * CT) / ocpg, ng = CT / ocpg;   // one wave-instruction of global atomics per workgroup
    if (tid < 2 * ng)
      atomicAdd((unsigned long long*)&a.out_stats[(n * a.out_groups + g0) * 2 + tid], (unsigned long long)sstat[2 * g0 + tid]);
  }
}

template <int KS, int CK, int CT, int PXF>
int launch2_cfg(ConvArgs a, hipStream_t st) {
  using C = Cfg2<KS, CK, CT, PXF>;
  a.tiles_x = cdiv(a.Wo, C::TW2);
  a.tiles_y = cdiv(a.Ho, C::TH2);
  a.g_T = a.tiles_x * a.tiles_y;
  dim3 grid(a.N * a.tiles_x * a.tiles_y, a.Cout / CT);
  const bool res = a.res != nullptr;
  // mode 1 multiplies fp16 operands and needs the fp16 weight pack; fp16 storage with bf16-packed weights takes the
  // run-time-flag instantiation (operands converted to bf16 in the loader)
  const int fm = (a.w_f16 && a.in_f16 && a.out_f16 && (a.res_f16 || !res)) ? 1
               : (!a.in_f16 && !a.out_f16 && (!a.res_f16 || !res)) ? 2
               : (!a.in_f16 && !a.out_f16 && a.res_f16) ? 3 : 0;
  // compile-time specialisations the engine's launches hit (anything else: the run-time-flag instantiation)
  const bool fwd_silu = fm == 1 && a.prologue == PTI_PRO_GN_SILU && !a.gn_mode && !a.pool2;
  const bool fwd_plain = fm == 1 && a.prologue == PTI_PRO_NONE && !a.gn_mode && !a.pool2;
  const bool dgrad = fm == 2 && a.prologue == PTI_PRO_NONE && !a.gn_mode && !a.out_stats;
  const bool dgrad_gn = fm == 3 && a.prologue == PTI_PRO_NONE && a.gn_mode && !a.out_stats && !a.pool2;
  if (a.x2) {   // data gradient + GroupNorm backward with the GroupNorm backward of the layer above as its prologue
    if constexpr (KS == 3 && PXF == 4 && CT == 128) {
      if (!(fm == 3 && a.gn_mode && a.act_out && !a.out_stats && !a.pool2 && a.mode == PTI_CONV_S1 && !a.relu_out)) return 4;
      PTI_LAUNCH((conv_mfma2_kernel<KS, CK, CT, PXF, true, 3, PRO_GNB>), grid, dim3(256), 0, st, a);
      return 0;
    } else {
      return 4;
    }
  }
  if (a.relu_out && !(fwd_plain && !a.act_out)) return 3;   // only that kernel has the fused ReLU
  // forward GroupNorm+SiLU launches: the prologue is a compile-time constant only for the 2-workgroup/CU shapes; at
  // the 128-VGPR cap of the others it made the compiler interleave the SiLU chains and spill (32->32@256^2 +res+stats
  // 153 -> 184 us), so those keep the run-time prologue flag (but the compile-time formats).  (tried: 3 workgroups/CU
  // = 168 VGPRs for those launches, compile-time prologue + packed SiLU, no spills: 32->32 152 -> 146 us alone, but
  // the training step got 1.5 % slower -- fewer resident waves overlap worse with the weight-gradient stream)
  constexpr int FPRO = PXF == 4 ? PTI_PRO_GN_SILU : -1;
  // (GroupNorm WITHOUT SiLU + side output -- the decoder's conv_out on the zero-padded image tile, csrc/narrow_pad.hip --
  //  takes the same instantiation where its prologue is a run-time flag, i.e. on the 32/64-wide tiles)
  const bool fwd_gn_rt = fm == 1 && a.prologue == PTI_PRO_GN && !a.gn_mode && !a.pool2 && FPRO < 0;
  if constexpr (KS == 3) {   // the activated-input side output is a separate instantiation (3x3 only)
    if (a.act_out) {
      if (fwd_silu || fwd_gn_rt) PTI_LAUNCH((conv_mfma2_kernel<KS, CK, CT, PXF, true, 1, FPRO>), grid, dim3(256), 0, st, a);
      else if (a.w_f16) return 2;
      else PTI_LAUNCH((conv_mfma2_kernel<KS, CK, CT, PXF, true, 0, -1>), grid, dim3(256), 0, st, a);
      return 0;
    }
  } else if (a.act_out) {
    return 1;
  }
  if (fwd_silu) PTI_LAUNCH((conv_mfma2_kernel<KS, CK, CT, PXF, false, 1, FPRO>), grid, dim3(256), 0, st, a);
  else if (fwd_plain) PTI_LAUNCH((conv_mfma2_kernel<KS, CK, CT, PXF, false, 1, PTI_PRO_NONE>), grid, dim3(256), 0, st, a);
  // fp16 operands with any other prologue (GroupNorm without SiLU: the encoder's conv_out on the padded latent tile):
  // the fp16-operand kernel with the run-time prologue flag.  (It used to fall through to the run-time-FORMAT
  // instantiation below, which multiplies bf16 operands -- with fp16-packed weights that is garbage.)
  else if (fm == 1 && !a.gn_mode && !a.pool2) PTI_LAUNCH((conv_mfma2_kernel<KS, CK, CT, PXF, false, 1, -1>), grid, dim3(256), 0, st, a);
  else if (a.w_f16) return 2;   // no fp16-operand kernel for this launch: refuse instead of mis-reading the weights
  else if (dgrad) PTI_LAUNCH((conv_mfma2_kernel<KS, CK, CT, PXF, false, 2, PTI_PRO_NONE>), grid, dim3(256), 0, st, a);
  else if (dgrad_gn) PTI_LAUNCH((conv_mfma2_kernel<KS, CK, CT, PXF, false, 3, PTI_PRO_NONE>), grid, dim3(256), 0, st, a);
  else PTI_LAUNCH((conv_mfma2_kernel<KS, CK, CT, PXF, false, 0, -1>), grid, dim3(256), 0, st, a);
  return 0;
}
template <int KS>
int launch2(const ConvArgs& a, int ck, int ct, hipStream_t st) {
  if (ct == 128) {
    if (ck == 128) return launch2_cfg<KS, 128, 128, 4>(a, st);
    if (ck == 64) return launch2_cfg<KS, 64, 128, 4>(a, st);
    return launch2_cfg<KS, 32, 128, 4>(a, st);
  }
  if (ct == 64) {
    if (ck == 64) return launch2_cfg<KS, 64, 64, 2>(a, st);
    return launch2_cfg<KS, 32, 64, 2>(a, st);
  }
  return launch2_cfg<KS, 32, 32, 2>(a, st);
}

// ---------------------------------------------------------------------------------------------
// weight packing: fp32 OIHW -> bf16 [cout tile][cin chunk][k-block][nt][lane][8]
// ---------------------------------------------------------------------------------------------
struct PackArgs {
  const float* src[4];
  int nsrc;
  bf16* dst;
  int cout_l, cin_l;   // logical (as seen by the consuming conv) channel counts
  int cout_o, cin_o;   // original weight dims (per source)
  int ks, ck, cout_tile, flip;
  int f16;   // store IEEE fp16 instead of bf16 (operands of the fp16 forward MFMA)
  long long total;
};

// one thread packs the 8 consecutive bf16 of one lane's fragment piece (one 16-byte store); 32-bit index math
__device__ __forceinline__ void pack_eight(const PackArgs& p, int e8) {
  const int NT = p.cout_tile / 32, KPC = p.ck / 16, KBC = p.ks * p.ks * KPC, nch = p.cin_l / p.ck;
  int r = e8;
  const int lane = r % 64; r /= 64;
  const int nt = r % NT; r /= NT;
  const int kb = r % KBC; r /= KBC;
  const int chunk = r % nch;
  const int ct = r / nch;
  const int co = ct * p.cout_tile + nt * 32 + (lane & 31);
  const int tap = kb / KPC, kc = kb % KPC;
  const int ci0 = chunk * p.ck + kc * 16 + 8 * (lane >> 5);
  const int kk = p.ks * p.ks;
  const float* src;
  size_t stride;
  if (!p.flip) {
    const int s_ = co / p.cout_o, cs = co % p.cout_o;
    src = p.src[s_] + ((size_t)cs * p.cin_o + ci0) * kk + tap;
    stride = kk;
  } else {  // logical co is an original input channel, logical ci an original output channel
    src = p.src[0] + ((size_t)ci0 * p.cin_o + co) * kk + (kk - 1 - tap);
    stride = (size_t)p.cin_o * kk;
  }
  float f[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = src[j * stride];
  *(u32x4*)(p.dst + (size_t)e8 * 8) = p.f16 ? pack8f(f, true) : pack8(f);
}

__global__ void pack_weights_kernel(PackArgs p) {
  if (p.f16) fp16_saturate_on();
  const long long e8 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e8 * 8 < p.total) pack_eight(p, (int)e8);
}

// batched form: table[i] describes one weight, blk_first[i] its first block; blocks are 256 threads x 8 elements
__global__ void pack_weights_batched_kernel(const PackArgs* __restrict__ table, const int* __restrict__ blk_first, int n) {
  int lo = 0, hi = n - 1;
  const int b = blockIdx.x;
  while (lo < hi) {  // last entry with blk_first <= b
    const int mid = (lo + hi + 1) >> 1;
    if (blk_first[mid] <= b) lo = mid; else hi = mid - 1;
  }
  const PackArgs p = table[lo];
  if (p.f16) fp16_saturate_on();   // p is block-uniform
  const int e8 = (b - blk_first[lo]) * 256 + threadIdx.x;
  if ((long long)e8 * 8 < p.total) pack_eight(p, e8);
}

template <int KS, int S, int CK, int COUT_TILE>
int launch_cfg(const ConvArgs& a, hipStream_t st) {
  dim3 grid(a.N * a.tiles_x * a.tiles_y, a.Cout / COUT_TILE);
  if (a.w_f16) PTI_LAUNCH((conv_mfma_kernel<KS, S, CK, COUT_TILE, true>), grid, dim3(256), 0, st, a);
  else PTI_LAUNCH((conv_mfma_kernel<KS, S, CK, COUT_TILE, false>), grid, dim3(256), 0, st, a);
  return 0;
}

template <int KS, int S, int CK>
int launch_ct(const ConvArgs& a, int cout_tile, hipStream_t st) {
  switch (cout_tile) {
    case 128: return launch_cfg<KS, S, CK, 128>(a, st);
    case 64: return launch_cfg<KS, S, CK, 64>(a, st);
    default: return launch_cfg<KS, S, CK, 32>(a, st);
  }
}

template <int KS, int S>
int launch_ck(const ConvArgs& a, int ck, int cout_tile, hipStream_t st) {
  static_assert(S == 2, "the v1 kernel is kept for the stride-2 gather only");
  switch (ck) {
    case 64: return launch_ct<KS, S, 64>(a, cout_tile, st);
    default: return launch_ct<KS, S, 32>(a, cout_tile, st);
  }
}

}  // namespace

extern "C" int64_t pti_conv_packed_bytes(int cout, int cin, int ksize, int mode) {
  (void)mode;
  if (cout <= 0 || cin <= 0 || cout % 32 || cin % 32 || (ksize != 1 && ksize != 3)) return 0;
  return (int64_t)2 * cout * cin * ksize * ksize;
}

extern "C" int pti_conv_pack_weights(const float* const* w, int nsrc, void* packed, int cout, int cin,
                                     int ksize, int mode, int transpose_flip, int w_f16, pti_stream_t s) {
  if (!w || !packed || nsrc < 1 || nsrc > 4) PTI_FAIL(PTI_EINVAL, "pack_weights: bad pointer/nsrc");
  if (cout % 32 || cin % 32 || (ksize != 1 && ksize != 3))
    PTI_FAIL(PTI_EUNSUPPORTED, "pack_weights: cout=%d cin=%d k=%d need multiples of 32, k in {1,3}", cout, cin, ksize);
  if (transpose_flip && nsrc != 1) PTI_FAIL(PTI_EINVAL, "pack_weights: transpose_flip needs nsrc=1");
  PackArgs p;
  for (int i = 0; i < 4; ++i) p.src[i] = i < nsrc ? w[i] : nullptr;
  p.nsrc = nsrc;
  p.dst = (bf16*)packed;
  p.cout_o = cout;
  p.cin_o = cin;
  p.cout_l = transpose_flip ? cin : cout * nsrc;
  p.cin_l = transpose_flip ? cout : cin;
  p.ks = ksize;
  p.flip = transpose_flip;
  p.f16 = w_f16 ? 1 : 0;
  p.cout_tile = pick_cout_tile(p.cout_l);
  p.ck = (mode == PTI_CONV_S2PAD) ? pick_ck(p.cin_l, true) : pick_ck2(p.cin_l, p.cout_tile);
  p.total = (long long)p.cout_l * p.cin_l * ksize * ksize;
  const int blocks = (int)((p.total + 2047) / 2048);
  PTI_LAUNCH(pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, p);
  PTI_CHECK_LAUNCH("pack_weights");
  return PTI_OK;
}

static int fill_pack(PackArgs& p, const float* w, void* packed, int cout, int cin, int ksize, int mode, int flip, int f16) {
  if (cout % 32 || cin % 32 || (ksize != 1 && ksize != 3)) return -1;
  for (int i = 0; i < 4; ++i) p.src[i] = nullptr;
  p.src[0] = w;
  p.nsrc = 1;
  p.dst = (bf16*)packed;
  p.cout_o = cout; p.cin_o = cin;
  p.cout_l = flip ? cin : cout;
  p.cin_l = flip ? cout : cin;
  p.ks = ksize; p.flip = flip; p.f16 = f16 ? 1 : 0;
  p.cout_tile = pick_cout_tile(p.cout_l);
  p.ck = (mode == PTI_CONV_S2PAD) ? pick_ck(p.cin_l, true) : pick_ck2(p.cin_l, p.cout_tile);
  p.total = (long long)p.cout_l * p.cin_l * ksize * ksize;
  return 0;
}

// Host-side table builder + single launch: entries i = 0..n-1 (w[i] fp32 [cout,cin,k,k] contiguous, possibly a
// fused [3C,C] view).  table_dev / blk_dev: device scratch of n*sizeof(pti_pack_entry_t) (96 B) / n*4 bytes
// that the CALLER filled through pti_conv_pack_table_fill (host) + its own H2D copy.
extern "C" int pti_conv_pack_entry_bytes(void) { return (int)sizeof(PackArgs); }
extern "C" int pti_conv_pack_table_fill(void* host_entry, const float* w, void* packed, int cout, int cin, int ksize,
                                        int mode, int transpose_flip, int w_f16, int64_t* nblocks) {
  if (!host_entry || !w || !packed || !nblocks) PTI_FAIL(PTI_EINVAL, "pack_table_fill: null pointer");
  PackArgs p;
  if (fill_pack(p, w, packed, cout, cin, ksize, mode, transpose_flip, w_f16))
    PTI_FAIL(PTI_EUNSUPPORTED, "pack_table_fill: cout=%d cin=%d k=%d", cout, cin, ksize);
  *(PackArgs*)host_entry = p;
  *nblocks = (p.total + 2047) / 2048;
  return PTI_OK;
}
extern "C" int pti_conv_pack_weights_batched(const void* table_dev, const int* blk_first_dev, int n, int total_blocks,
                                             pti_stream_t s) {
  if (!table_dev || !blk_first_dev || n <= 0 || total_blocks <= 0) PTI_FAIL(PTI_EINVAL, "pack_weights_batched: bad args");
  PTI_LAUNCH(pack_weights_batched_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)s,
                     (const PackArgs*)table_dev, blk_first_dev, n);
  PTI_CHECK_LAUNCH("pack_weights_batched");
  return PTI_OK;
}

struct GnBwdFuse {
  int mode; const int64_t* stats; const float* gamma; const float* beta; float* sums;
  // chained form: the launch's input is (g, x2) of the GroupNorm ABOVE and the staged operand its backward (PRO_GNB)
  const void* x2 = nullptr; int x2_f16 = 0; const int64_t* p_stats = nullptr; const float* p_gamma = nullptr; const float* p_sums = nullptr;
};

static int conv2d_mfma_impl(const void* x, const void* w_packed, const float* bias, const int64_t* in_stats,
                            const float* gamma, const float* beta, const void* residual, void* y,
                            int64_t* out_stats, const pti_conv_desc* d, const GnBwdFuse* gf, void* act_out,
                            pti_stream_t s) {
  if (!x || !w_packed || !y || !d) PTI_FAIL(PTI_EINVAL, "conv2d_mfma: null pointer");
  const bool chain = gf && gf->x2;
  if (act_out && !chain && (d->mode != PTI_CONV_S1 || d->prologue == PTI_PRO_NONE || d->ksize != 3))
    PTI_FAIL(PTI_EUNSUPPORTED, "conv2d_mfma: the activated-input side output needs a 3x3 PTI_CONV_S1 launch with a GroupNorm prologue");
  if (d->cin % 32 || d->cout % 32 || d->cin <= 0 || d->cout <= 0)
    PTI_FAIL(PTI_EUNSUPPORTED, "conv2d_mfma: cin=%d cout=%d must be positive multiples of 32", d->cin, d->cout);
  if (d->ksize != 1 && d->ksize != 3) PTI_FAIL(PTI_EUNSUPPORTED, "conv2d_mfma: ksize %d", d->ksize);
  if (d->ksize == 1 && d->mode != PTI_CONV_S1) PTI_FAIL(PTI_EUNSUPPORTED, "conv2d_mfma: 1x1 supports PTI_CONV_S1 only");
  if (d->n <= 0 || d->h <= 0 || d->w <= 0) PTI_FAIL(PTI_EINVAL, "conv2d_mfma: bad dims");
  int eho, ewo;
  switch (d->mode) {
    case PTI_CONV_S1: eho = d->h; ewo = d->w; break;
    case PTI_CONV_S2PAD: eho = (d->h + 1 - 3) / 2 + 1; ewo = (d->w + 1 - 3) / 2 + 1; break;
    case PTI_CONV_UP2: case PTI_CONV_ZINS: eho = 2 * d->h; ewo = 2 * d->w; break;
    default: PTI_FAIL(PTI_EUNSUPPORTED, "conv2d_mfma: mode %d", d->mode);
  }
  if (d->ho != eho || d->wo != ewo)
    PTI_FAIL(PTI_EINVAL, "conv2d_mfma: output %dx%d does not match mode %d on %dx%d (want %dx%d)", d->ho, d->wo,
             d->mode, d->h, d->w, eho, ewo);
  if (d->prologue != PTI_PRO_NONE) {
    if (!in_stats || !gamma || !beta || d->groups <= 0 || d->cin % d->groups)
      PTI_FAIL(PTI_EINVAL, "conv2d_mfma: prologue needs stats/gamma/beta and groups | cin");
    if (d->groups > 32) PTI_FAIL(PTI_EUNSUPPORTED, "conv2d_mfma: prologue supports at most 32 groups");
  }
  if (d->add_residual && !residual) PTI_FAIL(PTI_EINVAL, "conv2d_mfma: add_residual without residual");
  if (d->accum_stats) {
    if (!out_stats || d->out_groups <= 0 || d->cout % d->out_groups) PTI_FAIL(PTI_EINVAL, "conv2d_mfma: bad out stats");
    const int ocpg = d->cout / d->out_groups;
    if (ocpg != 2 && (ocpg % 4 != 0 || ocpg > 32)) PTI_FAIL(PTI_EUNSUPPORTED, "conv2d_mfma: channels/group %d for fused stats", ocpg);
    if (d->out_groups > 64) PTI_FAIL(PTI_EUNSUPPORTED, "conv2d_mfma: fused statistics support at most 64 groups");
  }
  ConvArgs a;
  a.x = (const bf16*)x; a.w = (const unsigned char*)w_packed; a.bias = bias; a.in_stats = (const stat_t*)in_stats;
  a.gamma = gamma; a.beta = beta; a.res = d->add_residual ? (const bf16*)residual : nullptr; a.y = (bf16*)y;
  a.out_stats = d->accum_stats ? (stat_t*)out_stats : nullptr;
  a.N = d->n; a.H = d->h; a.W = d->w; a.Cin = d->cin; a.Ho = d->ho; a.Wo = d->wo; a.Cout = d->cout;
  a.mode = d->mode; a.prologue = d->prologue; a.groups = d->groups; a.out_groups = d->out_groups;
  a.gn_mode = 0; a.g_groups = 0; a.g_inv_cnt = 0.f; a.g_eps = 0.f;
  a.g_stats = nullptr; a.g_gamma = a.g_beta = nullptr; a.g_sums = nullptr; a.g_T = 0;
  a.act_out = (bf16*)act_out;
  a.in_f16 = d->in_f16; a.res_f16 = d->res_f16; a.out_f16 = d->out_f16;
  a.pool2 = d->pool2x2_out;
  a.w_f16 = d->w_f16;
  a.relu_out = d->relu_out;
  if (a.relu_out && (gf || d->mode == PTI_CONV_S2PAD))
    PTI_FAIL(PTI_EUNSUPPORTED, "conv2d_mfma: relu_out is for plain fp16 forward launches (stride-1 gather, no prologue)");
  if (a.w_f16 && !(d->in_f16 && d->out_f16 && (d->res_f16 || !d->add_residual)) )
    PTI_FAIL(PTI_EUNSUPPORTED, "conv2d_mfma: fp16-packed weights need fp16 input, output and residual (the forward convs)");
  if (a.w_f16 && (gf || d->pool2x2_out))
    PTI_FAIL(PTI_EUNSUPPORTED, "conv2d_mfma: fp16-packed weights are for forward launches only");
  if (a.pool2 && (d->mode == PTI_CONV_S2PAD || d->accum_stats || gf || (d->ho & 1) || (d->wo & 1)))
    PTI_FAIL(PTI_EUNSUPPORTED, "conv2d_mfma: pool2x2_out needs a stride-1 gather, even output size, no fused statistics");
  a.eps = d->eps;
  a.inv_cnt = d->prologue != PTI_PRO_NONE ? 1.0f / ((float)(d->cin / d->groups) * (float)d->h * (float)d->w) : 0.f;
  if (gf) {
    if (d->groups > 32) PTI_FAIL(PTI_EUNSUPPORTED, "conv2d_mfma_gnbwd: at most 32 groups");
    a.gn_mode = gf->mode; a.g_groups = d->groups; a.g_eps = d->eps;
    a.g_inv_cnt = 1.0f / ((float)(d->cout / d->groups) * (float)d->ho * (float)d->wo);
    a.g_stats = (const stat_t*)gf->stats; a.g_gamma = gf->gamma; a.g_beta = gf->beta; a.g_sums = gf->sums;
  }
  a.x2 = nullptr; a.p_sums = nullptr; a.x2_f16 = 0;
  if (chain) {
    if (!gf->p_stats || !gf->p_gamma || !gf->p_sums || !act_out || d->mode != PTI_CONV_S1 || d->ksize != 3 ||
        d->cin % d->groups || d->cin / d->groups < 8)
      PTI_FAIL(PTI_EINVAL, "conv2d_mfma_gnbwd_chain: bad arguments (3x3 stride-1, >= 8 channels per group)");
    a.x2 = (const bf16*)gf->x2; a.x2_f16 = gf->x2_f16; a.p_sums = gf->p_sums;
    a.in_stats = (const stat_t*)gf->p_stats; a.gamma = gf->p_gamma; a.groups = d->groups;
    a.inv_cnt = 1.0f / ((float)(d->cin / d->groups) * (float)d->h * (float)d->w);
  }
  if ((long long)d->n * d->ho * d->wo * d->cout * 2 >= (1ll << 31))
    PTI_FAIL(PTI_EUNSUPPORTED, "conv2d_mfma: output tensor of %lld bytes (the stores use 32-bit offsets: < 2 GiB)",
             (long long)d->n * d->ho * d->wo * d->cout * 2);
  if (d->mode != PTI_CONV_S2PAD && (long long)d->n * d->h * d->w * d->cin * 2 >= (1ll << 31))
    PTI_FAIL(PTI_EUNSUPPORTED, "conv2d_mfma: input tensor of %lld bytes (the halo loads use 32-bit offsets: < 2 GiB)",
             (long long)d->n * d->h * d->w * d->cin * 2);
  a.tiles_x = cdiv(d->wo, TW); a.tiles_y = cdiv(d->ho, TH);
  const int cout_tile = pick_cout_tile(d->cout);
  int rc, ck;
  if (d->mode == PTI_CONV_S2PAD) {
    ck = pick_ck(d->cin, true);
    rc = launch_ck<3, 2>(a, ck, cout_tile, (hipStream_t)s);
  } else {
    ck = pick_ck2(d->cin, cout_tile);
    // 128 -> 128 3x3: the weight-stationary persistent kernel (conv_ws.hip) takes the launches it covers (rc 1 = not one
    // of them: the v2 kernel below)
    rc = (d->ksize == 3 && d->cin == 128 && d->cout == 128 && !a.x2) ? launch_conv_ws128(a, (hipStream_t)s) : 1;
    if (rc == 1)
      rc = d->ksize == 1 ? launch2<1>(a, ck, cout_tile, (hipStream_t)s) : launch2<3>(a, ck, cout_tile, (hipStream_t)s);
  }
  if (rc == 3) PTI_FAIL(PTI_EUNSUPPORTED, "conv2d_mfma: relu_out needs a plain fp16 forward launch (w_f16, fp16 in/out, no prologue, no side output)");
  if (rc == 2) PTI_FAIL(PTI_EUNSUPPORTED, "conv2d_mfma: no fp16-operand kernel for this launch (w_f16 with this prologue / epilogue)");
  if (rc == 4) PTI_FAIL(PTI_EUNSUPPORTED, "conv2d_mfma_gnbwd_chain: only 3x3 launches on 128-wide input and output tiles with an fp16 GroupNorm input");
  if (rc != 0) PTI_FAIL(PTI_EUNSUPPORTED, "conv2d_mfma: no kernel for ck=%d cout_tile=%d", ck, cout_tile);
  PTI_CHECK_LAUNCH("conv2d_mfma");
  return PTI_OK;
}

extern "C" int pti_conv2d_mfma(const void* x, const void* w_packed, const float* bias, const int64_t* in_stats,
                               const float* gamma, const float* beta, const void* residual, void* y,
                               int64_t* out_stats, const pti_conv_desc* d, pti_stream_t s) {
  return conv2d_mfma_impl(x, w_packed, bias, in_stats, gamma, beta, residual, y, out_stats, d, nullptr, nullptr, s);
}

extern "C" int pti_conv2d_mfma_saveact(const void* x, const void* w_packed, const float* bias, const int64_t* in_stats,
                                       const float* gamma, const float* beta, const void* residual, void* y,
                                       int64_t* out_stats, void* act_out, const pti_conv_desc* d, pti_stream_t s) {
  if (!act_out) PTI_FAIL(PTI_EINVAL, "conv2d_mfma_saveact: null act_out");
  return conv2d_mfma_impl(x, w_packed, bias, in_stats, gamma, beta, residual, y, out_stats, d, nullptr, act_out, s);
}

// pixel tiles per sample of the launch pti_conv2d_mfma_gnbwd(d) makes: the partial-sum buffer holds
// n * cout * 2 * tiles floats
extern "C" int pti_conv_gnbwd_tiles(const pti_conv_desc* d) {
  if (!d || d->cout % 32 || d->ho <= 0 || d->wo <= 0) return 0;
  const int ct = pick_cout_tile(d->cout);
  const int th = ct == 128 ? 8 : (ct == 64 ? 8 : 16);   // Cfg2::TH2 = 2 * PXF * WM with PXF = 4 (ct 128) or 2
  return cdiv(d->ho, th) * cdiv(d->wo, 16);
}

// chained form of pti_conv2d_mfma_gnbwd: supported shapes (see launch2_cfg)
extern "C" int pti_conv_gnbwd_chain_supported(int cin, int cout, int ksize) {
  return ksize == 3 && cin % 128 == 0 && cout % 128 == 0;
}

extern "C" int pti_conv2d_mfma_gnbwd_chain(const void* g_in, const void* x_in, int x_in_f16, const int64_t* in_stats,
                                           const float* in_gamma, const float* in_sums, void* dx_in_out, const void* w_packed,
                                           const void* gx, const int64_t* gstats, const float* ggamma, const float* gbeta,
                                           void* dy_out, float* gsums, const pti_conv_desc* d, int silu, pti_stream_t s) {
  if (!g_in || !x_in || !in_stats || !in_gamma || !in_sums || !dx_in_out || !gx || !gstats || !ggamma || !gbeta || !gsums || !d)
    PTI_FAIL(PTI_EINVAL, "conv2d_mfma_gnbwd_chain: null pointer");
  if (d->mode != PTI_CONV_S1 || d->prologue != PTI_PRO_NONE || d->add_residual || d->accum_stats)
    PTI_FAIL(PTI_EUNSUPPORTED, "conv2d_mfma_gnbwd_chain: plain stride-1 data-gradient launches only");
  if (d->groups <= 0 || d->cout % d->groups || d->cin % d->groups) PTI_FAIL(PTI_EINVAL, "conv2d_mfma_gnbwd_chain: groups must divide cin and cout");
  GnBwdFuse gf{silu ? 2 : 1, gstats, ggamma, gbeta, gsums};
  gf.x2 = x_in; gf.x2_f16 = x_in_f16; gf.p_stats = in_stats; gf.p_gamma = in_gamma; gf.p_sums = in_sums;
  pti_conv_desc dd = *d;
  dd.add_residual = 1;   // the GN input rides the residual path into LDS
  return conv2d_mfma_impl(g_in, w_packed, nullptr, nullptr, nullptr, nullptr, gx, dy_out, nullptr, &dd, &gf, dx_in_out, s);
}

extern "C" int pti_conv2d_mfma_gnbwd(const void* dy_in, const void* w_packed, const void* gx, const int64_t* gstats,
                                     const float* ggamma, const float* gbeta, void* dy_out, float* gsums,
                                     const pti_conv_desc* d, int silu, pti_stream_t s) {
  if (!gx || !gstats || !ggamma || !gbeta || !gsums || !d) PTI_FAIL(PTI_EINVAL, "conv2d_mfma_gnbwd: null pointer");
  if (d->mode == PTI_CONV_S2PAD || d->prologue != PTI_PRO_NONE || d->add_residual || d->accum_stats)
    PTI_FAIL(PTI_EUNSUPPORTED, "conv2d_mfma_gnbwd: plain stride-1 / zero-insert data-gradient launches only");
  if (d->groups <= 0 || d->cout % d->groups) PTI_FAIL(PTI_EINVAL, "conv2d_mfma_gnbwd: groups must divide cout");
  GnBwdFuse gf{silu ? 2 : 1, gstats, ggamma, gbeta, gsums};
  pti_conv_desc dd = *d;
  dd.add_residual = 1;   // the GN input rides the residual path into LDS
  return conv2d_mfma_impl(dy_in, w_packed, nullptr, nullptr, nullptr, nullptr, gx, dy_out, nullptr, &dd, &gf, nullptr, s);
}
