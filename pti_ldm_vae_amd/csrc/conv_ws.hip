// Weight-stationary persistent 3x3 convolution for the 128 -> 128 channel layers (gfx950).
//
// Replaces nn.Conv2d(128, 128, 3, padding=1) inside MONAI's AEKLResBlock / Upsample (reference
// src/pti_ldm_vae/models/autoencoder.py:67-79; SURVEY.md App. B: 4096x128x1152 is the "primary >= 40 % MFMA" shape, 14
// launches per step at 64^2 and 32 at 32^2), its data gradient and the data gradient fused with the GroupNorm(+SiLU)
// backward reduction -- same prologues / epilogues / tile geometry as conv_mfma2_kernel<3,128,128,4,...> (conv_mfma.hip),
// so the two are interchangeable launch by launch, bit for bit (tests/test_gpu_conv_ws.py).  EXPERIMENTAL, OFF by default
// (PTI_CONV_WS=1): see "what was measured" at the end of the file.
//
// Why: profiles/r02_pmc_conv_pipes.txt has the v2 kernel at 23-33 % MFMA-busy on these shapes.  Each of its waves streams
// ALL 72 weight fragments of its 32 output channels from L2 for every 128-pixel tile: 1 KiB per 4 MFMAs per SIMD = 32
// B/clk per CU ~ 67 GB/s per CU at the full MFMA rate -- what one CU can pull from L2 (MI355X_MICROARCH.md: 66-73 GB/s per
// CU from a table shared by every workgroup).  So the weight stream alone caps it near 50 %, and a launch is only 1-2
// rounds of workgroups that are all in the same phase (stage, MFMA, store), so nothing overlaps.
//
// Here: ONE workgroup (4 waves, one per SIMD, up to 512 registers each) per CU, persistent over pixel tiles.  Wave w keeps
// the 72 weight fragments of output channels 32w..32w+31 (all 9 taps x 128 input channels = 288 VGPRs) in registers for
// the whole launch: the weights are read from L2 once per CU per launch instead of once per tile, and the main loop is
// LDS reads + MFMAs only.  The halo tile is double-buffered in LDS: tile t+1's global loads are issued before tile t's
// MFMA loop and its GroupNorm(+SiLU) transform + LDS writes are spread between the tap groups of that loop.
// LDS: 2 x 45 KiB halo + 34 KiB output-transpose tile + 32 KiB residual tile (LDS-DMA) + tables = 157.25 KiB.
#include <type_traits>
#include <utility>

#include "conv_common.h"

namespace {
using namespace pti_conv;

using WC = Cfg2<3, 128, 128, 4>;
constexpr int WS_HALO = WC::HALO_BYTES;                    // 10 x 18 pixels x 256 B
constexpr int WS_ET_OFF = 2 * WS_HALO;                     // output tile, transposed through LDS for 16-byte stores
constexpr int WS_STAT_OFF = WS_ET_OFF + WC::EPI_BYTES;     // 1 KiB of group sums + 256 B float statistics table
constexpr int WS_RT_OFF = WS_STAT_OFF + 1024 + 256;        // residual tile / GroupNorm input of the fused backward
constexpr int WS_DUMMY_OFF = WS_RT_OFF + WC::RT_BYTES;     // 1 KiB sink for masked-off halo pieces (keeps the staging code branch-free)
constexpr int WS_SCT_OFF = WS_DUMMY_OFF + 1024;           // GroupNorm scale / shift of the tile being staged: [64 channel pairs]{sc0, sc1, sh0, sh1}
constexpr int WS_AUX_OFF = WS_SCT_OFF + 1024;             // 512 B: bias[128] (forward / plain data gradient) or {mean, rstd}[groups] (fused GN backward)
constexpr int WS_LDS = WS_AUX_OFF + 512;
static_assert(WS_LDS <= 160 * 1024, "LDS budget");
static_assert(WC::KBC == 72 && WC::HITERS == 12 && WC::EITERS == 8 && WC::MPX == 128 && WC::HW == 18, "tile geometry");

// Staging schedule of the NEXT tile's 12 halo pieces per thread (16 B = 8 channels of one halo pixel) over the 72 k-blocks
// of the current tile's MFMA loop, through FOUR rolling register slots (16 VGPRs; piece p lives in slot p & 3).  A k-block
// is 4 MFMAs = 128 cycles with ~96 cycles of VALU issue beside them; one DWORD (2 channels) of a piece costs ~16 VALU
// instructions (4 of them transcendental) ~ 80 cycles, so exactly one dword slice is attached to a k-block:
//   before the loop: load pieces 0..3 | k-blocks  8..23: slices of pieces 0..3 | 32..47: pieces 4..7 | 56..71: pieces 8..11
//   piece p + 4 is loaded in the k-block after piece p's last slice (>= 16 k-blocks ~ 2000 cycles before its first slice)
struct WsWork { int piece, dword, issue; };
__device__ constexpr WsWork ws_work(int kb) {
  WsWork w{-1, -1, -1};
  const int g = kb >> 3;
  if (g == 1 || g == 2) w.piece = (kb - 8) >> 2;
  else if (g == 4 || g == 5) w.piece = 4 + ((kb - 32) >> 2);
  else if (g == 7 || g == 8) w.piece = 8 + ((kb - 56) >> 2);
  if (w.piece >= 0) w.dword = kb & 3;
  if (kb >= 12 && kb <= 24 && (kb & 3) == 0) w.issue = 4 + ((kb - 12) >> 2);
  if (kb >= 36 && kb <= 48 && (kb & 3) == 0) w.issue = 8 + ((kb - 36) >> 2);
  return w;
}
static_assert(ws_work(8).piece == 0 && ws_work(23).piece == 3 && ws_work(23).dword == 3 && ws_work(71).piece == 11, "slices");
static_assert(ws_work(12).issue == 4 && ws_work(24).issue == 7 && ws_work(36).issue == 8 && ws_work(48).issue == 11, "issues");
constexpr int WS_NA = 48;   // weight fragments pinned in AGPRs (192 registers; + 64 accumulators = the whole AGPR file)

template <int... I, class F>
__device__ __forceinline__ void ws_for(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }

struct Tile { int n, oy0, ox0, ty, tx; };

#ifdef WS_STAMPS
// Diagnostic build only (never the shipped library): s_memtime stamps of workgroup 0 .. 7, wave 0, first 12 tiles:
// [wg][tile][5] = tile start, MFMA loop start, loop end (before barrier a), after barrier (b), tile end.
__device__ long long pti_ws_stamps[8 * 12 * 5];
#define WS_STAMP(k) do { if (blockIdx.x < 8 && tidv == 0 && stamp_tile < 12) pti_ws_stamps[(blockIdx.x * 12 + stamp_tile) * 5 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WS_STAMP(k) do {} while (0)
#endif

// FM: 1 = fp16 in / residual / out, fp16 MFMA operands (forward); 2 = all bf16 (plain data gradient); 3 = bf16 in / out,
// fp16 "residual" = GroupNorm input of the fused GroupNorm(+SiLU)-backward epilogue.  PRO: PTI_PRO_NONE / _GN / _GN_SILU.
template <int FM, int PRO, bool SAVE>
__global__ __launch_bounds__(256, 1) void conv_ws128_kernel(ConvArgs a, int ntiles) {
  using C = WC;
  constexpr bool OPH = FM == 1, in_f16 = FM == 1, res_f16 = (FM == 1 || FM == 3), out_f16 = FM == 1, gn_on = FM == 3;
  constexpr int CT = 128;
  if (out_f16) fp16_saturate_on();
  __shared__ __attribute__((aligned(16))) unsigned char smem[WS_LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);   // this wave's 32-wide output-channel fragment
  const int hsel = lane >> 5, j = lane & 31, txl = lane & 15;
  const int pad_lo = (a.mode == PTI_CONV_ZINS) ? 2 : 1;
  const bool twox = (a.mode == PTI_CONV_UP2) || (a.mode == PTI_CONV_ZINS);
  const int VH = twox ? 2 * a.H : a.H, VW = twox ? 2 * a.W : a.W;

  // ---- weights: [kb][nt][lane][8] of cout tile 0 / cin chunk 0 -> 72 fragments in registers, once per launch ----
  bf16x8 wreg[72];
  {
    const unsigned char* wl = a.w + (size_t)wn * 1024 + lane * 16;
#pragma unroll
    for (int kb = 0; kb < 72; ++kb) wreg[kb] = *(const bf16x8*)(wl + kb * 4096);
    // the first WS_NA fragments live in AGPRs (the MFMA reads its A operand from there directly); without the pin the
    // register allocator keeps them all in VGPRs and spills
#pragma unroll
    for (int kb = 0; kb < WS_NA; ++kb) asm volatile("" : "+a"(wreg[kb]));
  }

  auto decode = [&](int t) -> Tile {
    Tile r;
    r.tx = t % a.tiles_x;
    t /= a.tiles_x;
    r.ty = t % a.tiles_y;
    r.n = t / a.tiles_y;
    r.oy0 = r.ty * C::TH2;
    r.ox0 = r.tx * C::TW2;
    return r;
  };

  // ---- halo loader state (constant per thread): piece = 8 channels lc*8.. of halo pixel lp0 + 16*it ----
  int lc = tid & 15, lp0 = tid >> 4;       // (laundered per tile, see the tile loop: keeps derived addresses out of registers)
  int tidv = tid, lanev = lane;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16*>(a.x), 0, (unsigned)a.N * (unsigned)a.H * (unsigned)a.W * 128u * 2u, 0x00020000);
  const unsigned rowb = (unsigned)a.W * 256u;
  int gshift = 0;
  if constexpr (PRO != PTI_PRO_NONE) gshift = __builtin_ctz(128 / a.groups);     // channels per group: a power of two (host check)
  u32x4 raw[4];            // rolling slots: piece p in slot p & 3
  unsigned okmask = 0;
  int ld_vy0 = 0, ld_vx0 = 0;      // tile being loaded: origin of its halo in the (virtual) input, sample base offset
  unsigned ld_base = 0;

  // per-tile part of the loader: origin / base of tile tl; and the loads for the GroupNorm scale / shift table of tl's
  // sample: lane l of EVERY wave fetches the statistics / gamma / beta of channel pair l (6 registers); tbl_write turns
  // them into {sc, sc, sh, sh} and stores entry l (the four waves store identical values: no branch, no race) -- once per
  // tile instead of every thread deriving its 8 channels into 16 registers.  A barrier separates tbl_write from the slices.
  float tb_sum = 0.f, tb_sq = 0.f;
  f32x2 tb_g = {0.f, 0.f}, tb_b = {0.f, 0.f};
  auto halo_begin = [&](const Tile& tl) {
    ld_vy0 = tl.oy0 - pad_lo;
    ld_vx0 = tl.ox0 - pad_lo;
    ld_base = ((unsigned)tl.n * a.H * a.W * 128u + lc * 8) * 2u;
    okmask = 0;
    if constexpr (PRO != PTI_PRO_NONE) {
      const int c0 = 2 * (tidv & 63);
      const stat_t* st = a.in_stats + ((size_t)tl.n * a.groups + (c0 >> gshift)) * 2;
      tb_sum = stat_f(st, 0);
      tb_sq = stat_f(st, 1);
      tb_g = *(const f32x2*)(a.gamma + c0);
      tb_b = *(const f32x2*)(a.beta + c0);
    }
  };
  auto tbl_write = [&]() {
    if constexpr (PRO != PTI_PRO_NONE) {
      const float mean = tb_sum * a.inv_cnt;
      const float rstd = __builtin_amdgcn_rsqf(fmaxf(tb_sq * a.inv_cnt - mean * mean, 0.f) + a.eps);   // (as gn_params)
      const float s0 = rstd * tb_g[0], s1 = rstd * tb_g[1];
      *(f32x4*)(smem + WS_SCT_OFF + (tidv & 63) * 16) = f32x4{s0, s1, tb_b[0] - mean * s0, tb_b[1] - mean * s1};
    }
  };
  // load piece `it` of the tile set up by halo_begin into its slot
  auto halo_issue = [&](int it) {
    const int p = lp0 + it * 16;
    const int hy = p / 18, hx = p - hy * 18;
    const int vy = ld_vy0 + hy, vx = ld_vx0 + hx;
    bool v = (p < C::NP) && (unsigned)vy < (unsigned)VH && (unsigned)vx < (unsigned)VW;
    int iy = vy, ix = vx;
    if (twox) {
      if (a.mode == PTI_CONV_ZINS) v = v && !((vy | vx) & 1);
      iy = vy >> 1;
      ix = vx >> 1;
    }
    okmask |= (v ? 1u : 0u) << it;
    const unsigned off = v ? ld_base + (unsigned)iy * rowb + (unsigned)ix * 256u : 0x80000000u;
    raw[it & 3] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0));
  };
  // LDS address of piece `it` in halo buffer hb; masked-off pieces (past the 180 halo pixels, or no next tile) go to a
  // per-lane sink slot instead of a branch: the staging code stays straight-line inside the MFMA loop
  auto halo_dst = [&](int it, unsigned char* hb, bool enable) -> unsigned char* {
    const int p = lp0 + it * 16;
    const int hy = p / 18, hx = p - hy * 18;
    return (enable && p < C::NP) ? hb + p * 256 + ((lc ^ (hx & 15)) << 4) : smem + WS_DUMMY_OFF + lanev * 16;
  };
  // dword d (channels 2d, 2d+1 of the piece) of piece `it`: GroupNorm affine (+SiLU) in place in its register slot
  // (out-of-image pieces keep their zeros through a select); the last dword writes the piece to LDS
  auto halo_slice = [&](int it, int d, unsigned char* hb, bool enable) {
    if constexpr (PRO != PTI_PRO_NONE) {
      const uint32_t w = raw[it & 3][d];
      float lo, hi;
      unpack2f(w, in_f16, lo, hi);
      const f32x4 ss = *(const f32x4*)(smem + WS_SCT_OFF + (lc * 4 + d) * 16);     // {sc, sc, sh, sh} of channels lc*8 + 2d, +1
      if constexpr (PRO == PTI_PRO_GN_SILU) {   // the v2 kernel's formula for this tile shape
        const f32x2 o = gn_silu2(f32x2{lo, hi}, f32x2{ss[0], ss[1]}, f32x2{ss[2], ss[3]});
        lo = o[0];
        hi = o[1];
      } else {
        lo = lo * ss[0] + ss[2];
        hi = hi * ss[1] + ss[3];
      }
      const uint32_t tw = pack2f(lo, hi, OPH);
      uint32_t out = ((okmask >> it) & 1u) ? tw : w;
      asm volatile("" : "+v"(out));     // anchor: the slice is computed HERE (IR passes otherwise sink all four slices of a
      raw[it & 3][d] = out;             // piece down to the k-block of its LDS write and the VALU work lumps up again)
    }
    if (d == 3) *(u32x4*)halo_dst(it, hb, enable) = raw[it & 3];
  };

  // ---- B (pixel) fragment addressing: lane j = pixel (row j>>4 of the fragment's 2 rows, column txl), k-half hsel ----
  int pb = ((j >> 4) * 18 + txl) * 256;
  int tk[3];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) tk[kw] = (hsel ^ ((txl + kw) & 15)) << 4;

  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
      a.y, 0, ((unsigned)a.N * (unsigned)a.Ho * (unsigned)a.Wo * 128u * 2u) >> (a.pool2 ? 2 : 0), 0x00020000);
  int epc = tid & 15, epp0 = tid >> 4;   // epilogue store: piece epc of pixel epp0 + 16*it

  if constexpr (!gn_on) {      // bias -> LDS once (a global load in the epilogue is an exposed L2 round trip per tile here)
    if (a.bias && tid < 128) reinterpret_cast<float*>(smem + WS_AUX_OFF)[tid] = a.bias[tid];
  }
  // ================= prologue: stage the first tile =================
  int t = blockIdx.x;
  int buf = 0;
  {
  const Tile cur = decode(t);
  halo_begin(cur);
  tbl_write();
  __syncthreads();       // the scale / shift table (and the bias table)
#pragma unroll
  for (int b4 = 0; b4 < 12; b4 += 4) {
#pragma unroll
    for (int it = b4; it < b4 + 4; ++it) halo_issue(it);
#pragma unroll
    for (int it = b4; it < b4 + 4; ++it)
#pragma unroll
      for (int d = 0; d < 4; ++d) halo_slice(it, d, smem, true);
  }
  __syncthreads();
  }

  int stamp_tile = 0;
  (void)stamp_tile;
  for (; t < ntiles; t += gridDim.x) {
    const Tile cur = decode(t);      // (re-derived per tile: carried across the loop the five scalars end up spilled)
    unsigned char* hcur = smem + buf * WS_HALO;
    unsigned char* hnxt = smem + (buf ^ 1) * WS_HALO;
    const int tn = t + gridDim.x;
    const bool has_next = tn < ntiles;     // workgroup-uniform
    const int n = cur.n, oy0 = cur.oy0, ox0 = cur.ox0;
    // Everything derived from these per-thread constants (12 halo-piece addresses, 24 B-fragment addresses, 8 + 8 store
    // offsets ...) is loop-invariant, and the compiler hoists it all out of the tile loop into registers this kernel does
    // not have (288 are weights): it then spills and reloads them around every use.  Making the seeds opaque per tile
    // keeps the 1-3 VALU instructions of each address next to its use instead.
    asm volatile("" : "+v"(lc), "+v"(lp0), "+v"(pb), "+v"(tk[0]), "+v"(tk[1]), "+v"(tk[2]), "+v"(epc), "+v"(epp0), "+v"(tidv), "+v"(lanev));
    const int hsel = lanev >> 5, j = lanev & 31;      // (shadow the kernel-scope copies inside the tile loop)

    WS_STAMP(0);
    // ---- tile start: residual tile by LDS-DMA, per-tile tables, next tile's loads ----
    reinterpret_cast<float*>(smem + WS_STAT_OFF)[tidv] = 0.f;
    if constexpr (gn_on) {
      // GroupNorm parameters of THIS tile's sample for the fused-backward epilogue: {scale, shift} per channel and
      // {mean, rstd} per group, computed once by 128 threads into LDS (barrier (a) lies between this and the epilogue);
      // per-lane global loads of gamma / beta / statistics in the epilogue were ~16 exposed L2 round trips per tile
      if (tidv < 128) {
        const int gsh = __builtin_ctz(128 / a.g_groups);
        const int g = tidv >> gsh;
        const float sum = stat_f(a.g_stats, (n * a.g_groups + g) * 2), sq = stat_f(a.g_stats, (n * a.g_groups + g) * 2 + 1);
        const float mean = sum * a.g_inv_cnt;
        const float rstd = __builtin_amdgcn_rsqf(fmaxf(sq * a.g_inv_cnt - mean * mean, 0.f) + a.g_eps);
        const float scv = rstd * a.g_gamma[tidv];
        *(f32x2*)(smem + WS_SCT_OFF + tidv * 8) = f32x2{scv, a.g_beta[tidv] - mean * scv};
        if ((tidv & ((1 << gsh) - 1)) == 0) *(f32x2*)(smem + WS_AUX_OFF + g * 8) = f32x2{mean, rstd};
      }
    }
    if (a.res) {
#pragma unroll
      for (int it = 0; it < C::EITERS; ++it) {
        const int slot = it * 256 + tidv;
        const int p = slot >> 4, c = slot & 15;
        const int oy = oy0 + (p >> 4), ox = ox0 + (p & 15);
        if (oy < a.Ho && ox < a.Wo) {
          const bf16* src = a.res + ((size_t)(n * a.Ho + oy) * a.Wo + ox) * 128 + ((c ^ ((p >> 2) & 15)) << 3);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                           (__attribute__((address_space(3))) void*)(smem + WS_RT_OFF + (it * 256 + wn * 64) * 16),
                                           16, 0, 0);
        }
      }
    }
    Tile nxt = cur;
    if (has_next) nxt = decode(tn);
    halo_begin(nxt);                       // (no next tile: the loads repeat this tile's and their LDS writes are masked off)
#pragma unroll
    for (int it = 0; it < 4; ++it) halo_issue(it);

    // side output act(GN(x)) of THIS tile (bf16, the weight gradient's operand): interior pixels of the staged halo, one
    // 16-byte piece per thread in each of the first 8 k-blocks of the MFMA loop
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(
        SAVE ? a.act_out : a.y, 0, (unsigned)a.N * (unsigned)a.H * (unsigned)a.W * 128u * 2u, 0x00020000);
    auto save_piece = [&](int it) {
      const int idx = tidv + it * 256;
      const int c8 = idx & 15, pi = idx >> 4;
      const int hy = (pi >> 4) + 1, hx = (pi & 15) + 1;
      const int vy = oy0 - 1 + hy, vx = ox0 - 1 + hx;
      u32x4 piece = *(const u32x4*)(hcur + (hy * 18 + hx) * 256 + ((c8 ^ (hx & 15)) << 4));
      if constexpr (OPH) {
        float f[8];
        unpack8f(piece, f, true);
        piece = pack8(f);
      }
      const unsigned off = (vy < a.H && vx < a.W) ? ((unsigned)((n * a.H + vy) * a.W + vx) * 128u + c8 * 8) * 2u : 0x80000000u;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(bu32x4, piece), srs, off, 0, 2 /* nt */);   // out of range: dropped
    };

    // ---- MFMA loop: 9 tap groups x 8 k-blocks x 4 pixel fragments; B fragments double-buffered in registers ----
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 b0[4], b1[4];
    auto bread = [&](bf16x8 (&dst)[4], int kb) {
      kb = kb < 72 ? kb : 71;
      const int tap = kb >> 3, kc = kb & 7;
      const int kh = tap / 3, kw = tap - 3 * kh;
      const unsigned char* bp = hcur + pb + (kh * 18 + kw) * 256 + ((kc * 32) ^ tk[kw]);
#pragma unroll
      for (int i = 0; i < 4; ++i) dst[i] = *(const bf16x8*)(bp + 2 * i * 18 * 256);
    };
    WS_STAMP(1);
    bread(b0, 0);
    __builtin_amdgcn_sched_barrier(0);
    // one k-block: the B reads of the NEXT k-block, 4 MFMAs, and this k-block's share of the staging / side-output work,
    // with the issue order pinned (the compiler otherwise sinks the B reads next to their MFMAs -- with one wave per SIMD
    // nothing else hides an LDS round trip -- and lumps the VALU work of a whole piece into one MFMA gap)
    auto kblock = [&](auto kc_) {
      constexpr int kb = decltype(kc_)::value;
      constexpr WsWork wk = ws_work(kb);
      if constexpr ((kb & 1) == 0) {
        bread(b1, kb + 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = mfma32<OPH>(wreg[kb], b0[i], acc[i]);
      } else {
        bread(b0, kb + 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = mfma32<OPH>(wreg[kb], b1[i], acc[i]);
      }
      // side output in the idle tap group 3 (k-blocks 24..31), NOT in group 0: the first slice (k-block 8) waits with
      // vmcnt(0) -- hipcc does not count past an LDS-DMA in flight -- and would sit there until the stores had drained
      if constexpr (SAVE && kb >= 24 && kb < 32) save_piece(kb - 24);
      // the next tile's scale / shift table: its loads were issued at tile start; written here, under the MFMAs, and made
      // visible by a RAW barrier two k-blocks later (__syncthreads() would add vmcnt(0) and drain the loads in flight)
      if constexpr (PRO != PTI_PRO_NONE && kb == 5) tbl_write();
      if constexpr (PRO != PTI_PRO_NONE && kb == 7) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      if constexpr (wk.piece >= 0) halo_slice(wk.piece, wk.dword, hnxt, has_next);
      if constexpr (wk.issue >= 0) halo_issue(wk.issue);
      constexpr int nv = (SAVE && kb >= 24 && kb < 32) ? 24 : ((wk.piece >= 0 && PRO != PTI_PRO_NONE) ? 16 : 0) + (wk.issue >= 0 ? 12 : 0)
                         + ((PRO != PTI_PRO_NONE && kb == 5) ? 10 : 0);
      constexpr int K = (nv + 3) / 4;
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);   // DS read x 4
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
        if constexpr (K > 0) __builtin_amdgcn_sched_group_barrier(0x002, K, 0);   // VALU
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    ws_for(std::make_integer_sequence<int, 72>{}, kblock);

    // ---- epilogue: bias, residual, 16-bit rounding, GroupNorm statistics / fused GroupNorm backward, coalesced store ----
    unsigned char* etile = smem + WS_ET_OFF;
    const bool do_stats = (FM == 2 || FM == 3) ? false : a.out_stats != nullptr;
    const int ocpg = do_stats ? 128 / a.out_groups : 1;
    WS_STAMP(2);
    __syncthreads();   // (a) every wave is done with hcur; the next halo and the residual tile are complete
    const unsigned char* rtile = smem + WS_RT_OFF;
    const int col0 = wn * 32 + 4 * hsel;
    float st1[4] = {0.f, 0.f, 0.f, 0.f}, st2[4] = {0.f, 0.f, 0.f, 0.f};
    float su1[4] = {0.f, 0.f, 0.f, 0.f}, su2[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (gn_on) {
      const int gsh_g = __builtin_ctz(128 / a.g_groups);
      float* gsm = reinterpret_cast<float*>(smem + WS_STAT_OFF);   // [CT][2]
      float L[32];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = col0 + 8 * q;
        float scv[4], shv[4], muv[4], rsv[4], l1[4] = {0.f, 0.f, 0.f, 0.f}, l2[4] = {0.f, 0.f, 0.f, 0.f};
        {
          const f32x4 t0 = *(const f32x4*)(smem + WS_SCT_OFF + col * 8), t1 = *(const f32x4*)(smem + WS_SCT_OFF + col * 8 + 16);
          scv[0] = t0[0]; shv[0] = t0[1]; scv[1] = t0[2]; shv[1] = t0[3];
          scv[2] = t1[0]; shv[2] = t1[1]; scv[3] = t1[2]; shv[3] = t1[3];
          const f32x2 m0 = *(const f32x2*)(smem + WS_AUX_OFF + (col >> gsh_g) * 8);
          const f32x2 m2 = *(const f32x2*)(smem + WS_AUX_OFF + ((col + 2) >> gsh_g) * 8);   // (2 channels per group: a second group)
          muv[0] = muv[1] = m0[0]; rsv[0] = rsv[1] = m0[1];
          muv[2] = muv[3] = m2[0]; rsv[2] = rsv[3] = m2[1];
        }
        float nmr[4];     // -mean * rstd: xhat = x * rstd + nmr
#pragma unroll
        for (int r = 0; r < 4; ++r) nmr[r] = -muv[r] * rsv[r];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int p = (2 * i + (j >> 4)) * 16 + (j & 15);
          const bool inb = (oy0 + (p >> 4) < a.Ho) && (ox0 + (p & 15) < a.Wo);
          unsigned char* ep = etile + p * C::EPITCH + col * 2;
          const u32x2 rr = *(const u32x2*)(rtile + p * 256 + ((((col >> 3) ^ (p >> 2)) & 15) << 4) + (col & 7) * 2);
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
      fold32<32>(L, j);
      gsm[col0 * 2 + 16 * (j >> 3) + (j & 7)] = L[0];
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int p = (2 * i + (j >> 4)) * 16 + (j & 15);
        const bool inb = (oy0 + (p >> 4) < a.Ho) && (ox0 + (p & 15) < a.Wo);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int col = col0 + 8 * q;
          float v0 = acc[i][4 * q + 0], v1 = acc[i][4 * q + 1], v2 = acc[i][4 * q + 2], v3 = acc[i][4 * q + 3];
          if (a.bias) {
            const f32x4 b = *(const f32x4*)(smem + WS_AUX_OFF + col * 4);
            v0 += b[0]; v1 += b[1]; v2 += b[2]; v3 += b[3];
          }
          unsigned char* ep = etile + p * C::EPITCH + col * 2;
          if (a.res) {
            const u32x2 rr = *(const u32x2*)(rtile + p * 256 + ((((col >> 3) ^ (p >> 2)) & 15) << 4) + (col & 7) * 2);
            float e0, e1, e2, e3;
            unpack2f(rr[0], res_f16, e0, e1);
            unpack2f(rr[1], res_f16, e2, e3);
            v0 += e0; v1 += e1; v2 += e2; v3 += e3;
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
      stat_t* sstat = reinterpret_cast<stat_t*>(smem + WS_STAT_OFF);
      if (ocpg >= 4) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) { v[2 * q] = st1[q]; v[2 * q + 1] = st2[q]; }
        fold32<8>(v, j);
        if ((j & 3) == 0) {
          const int idx = j >> 2, q = idx >> 1;
          const int g = (col0 + 8 * q) / ocpg;
          stat_add(&sstat[2 * g + (idx & 1)], v[0]);
        }
      } else {
        float v[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) { v[4 * q] = st1[q]; v[4 * q + 1] = st2[q]; v[4 * q + 2] = su1[q]; v[4 * q + 3] = su2[q]; }
        fold32<16>(v, j);
        if ((j & 1) == 0) {
          const int idx = j >> 1, q = idx >> 2;
          const int g = (col0 + 8 * q) / 2;
          stat_add(&sstat[2 * g + (idx & 3)], v[0]);
        }
      }
    }
    __syncthreads();   // (b) the output tile (and the group sums) are complete
    WS_STAMP(3);
    if ((FM == 2) && a.pool2) {
      // data gradient of conv(nearest-2x(x)): store the 2x2-sum-pooled tile (see conv_mfma2_kernel)
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int idx = tidv + it * 256;
        const int c8 = idx & 15, pp = idx >> 4;
        const int py = pp >> 3, px = pp & 7;
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
                                                 ((unsigned)((n * (a.Ho >> 1) + oy) * (a.Wo >> 1) + ox) * 128u + c8 * 8) * 2u, 0, 2);
        }
      }
    } else {
#pragma unroll
      for (int it = 0; it < C::EITERS; ++it) {
        const int p = epp0 + it * 16;
        const int oy = oy0 + (p >> 4), ox = ox0 + (p & 15);
        if (oy < a.Ho && ox < a.Wo)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(bu32x4, *(const u32x4*)(etile + p * C::EPITCH + epc * 16)), yrs,
                                                 ((unsigned)((n * a.Ho + oy) * a.Wo + ox) * 128u + epc * 8) * 2u, 0, 2);
      }
    }
    if constexpr (gn_on) {
      if (tidv < 2 * CT) {
        const float* gsm = reinterpret_cast<const float*>(smem + WS_STAT_OFF);
        a.g_sums[(((size_t)n * a.g_T + cur.ty * a.tiles_x + cur.tx) * 128) * 2 + tidv] = gsm[tidv];
      }
    } else if (do_stats) {
      const stat_t* sstat = reinterpret_cast<const stat_t*>(smem + WS_STAT_OFF);
      const int ng = 128 / ocpg;
      if (tidv < 2 * ng)
        atomicAdd((unsigned long long*)&a.out_stats[(n * a.out_groups) * 2 + tidv], (unsigned long long)sstat[tidv]);
    }
    __syncthreads();   // (c) etile / tables are free for the next tile
    WS_STAMP(4);
    ++stamp_tile;
    buf ^= 1;
  }
}

template <int FM, int PRO, bool SAVE>
int ws_launch(const ConvArgs& a, int ntiles, int grid, hipStream_t st) {
  PTI_LAUNCH((conv_ws128_kernel<FM, PRO, SAVE>), dim3(grid), dim3(256), 0, st, a, ntiles);
  return 0;
}

}  // namespace

namespace pti_conv {

int launch_conv_ws128(const ConvArgs& a0, hipStream_t st) {
  // OFF by default (PTI_CONV_WS=1 turns it on; both kernels give the same bits): measured on MI355X at batch 32 it ties
  // the v2 kernel on the launches without a prologue (plain data gradient 128^2: 140 vs 147 us) and LOSES 10-25 % on the
  // training step's forward (GN+SiLU + residual + statistics + side output) and fused-backward launches -- see the
  // "what was measured" block at the end of this file.  PTI_CONV_WS_MAX_WGS=n caps the grid (tests use it to drive many
  // tiles through each persistent workgroup on small tensors).  Both are read per launch.
  const char* e_on = getenv("PTI_CONV_WS");
  if (!e_on || atoi(e_on) == 0) return 1;
  const char* e_cap = getenv("PTI_CONV_WS_MAX_WGS");
  const int cap = e_cap ? atoi(e_cap) : 0;
  if (a0.Cin != 128 || a0.Cout != 128 || a0.relu_out) return 1;
  if (a0.mode != PTI_CONV_S1 && a0.mode != PTI_CONV_UP2 && a0.mode != PTI_CONV_ZINS) return 1;
  ConvArgs a = a0;
  a.tiles_x = cdiv(a.Wo, WC::TW2);
  a.tiles_y = cdiv(a.Ho, WC::TH2);
  a.g_T = a.tiles_x * a.tiles_y;
  const bool res = a.res != nullptr;
  const int fm = (a.w_f16 && a.in_f16 && a.out_f16 && (a.res_f16 || !res)) ? 1
               : (!a.in_f16 && !a.out_f16 && (!a.res_f16 || !res)) ? 2
               : (!a.in_f16 && !a.out_f16 && a.res_f16) ? 3 : 0;
  if (a.prologue != PTI_PRO_NONE) {
    const int cpg = 128 / a.groups;
    if (a.groups <= 0 || 128 % a.groups || (cpg & (cpg - 1)) || cpg < 2) return 1;
  }
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 1;
    cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const long long nt = (long long)a.N * a.tiles_x * a.tiles_y;
  if (nt <= 0 || nt > 0x7fffffff) return 1;
  const int ntiles = (int)nt;
  int grid = ntiles < cus ? ntiles : cus;
  if (cap > 0 && grid > cap) grid = cap;
  const bool fwd = fm == 1 && !a.gn_mode && !a.pool2;
  if (a.act_out) {
    if (fwd && a.prologue == PTI_PRO_GN_SILU && a.mode == PTI_CONV_S1) return ws_launch<1, PTI_PRO_GN_SILU, true>(a, ntiles, grid, st);
    return 1;
  }
  if (fwd && a.prologue == PTI_PRO_GN_SILU) return ws_launch<1, PTI_PRO_GN_SILU, false>(a, ntiles, grid, st);
  if (fwd && a.prologue == PTI_PRO_NONE) return ws_launch<1, PTI_PRO_NONE, false>(a, ntiles, grid, st);
  if (fwd && a.prologue == PTI_PRO_GN) return ws_launch<1, PTI_PRO_GN, false>(a, ntiles, grid, st);
  if (a.w_f16) return 1;
  if (fm == 3) {
    const int cpg = a.g_groups > 0 ? 128 / a.g_groups : 0;
    if (a.g_groups <= 0 || 128 % a.g_groups || (cpg & (cpg - 1)) || cpg < 2) return 1;
  }
  if (fm == 2 && a.prologue == PTI_PRO_NONE && !a.gn_mode && !a.out_stats) return ws_launch<2, PTI_PRO_NONE, false>(a, ntiles, grid, st);
  if (fm == 3 && a.prologue == PTI_PRO_NONE && a.gn_mode && !a.out_stats && !a.pool2) return ws_launch<3, PTI_PRO_NONE, false>(a, ntiles, grid, st);
  return 1;
}

#ifdef WS_STAMPS
extern "C" int pti_debug_ws_stamps(long long* dst_host) {
  return (int)hipMemcpyFromSymbol(dst_host, HIP_SYMBOL(pti_ws_stamps), sizeof(long long) * 8 * 12 * 5, 0, hipMemcpyDeviceToHost);
}
#endif

}  // namespace pti_conv

// ---- what was measured (round 3, MI355X, batch 32, tools/bench_conv_ws.py: interleaved A/B in one process) --------------
//                         v2 kernel          this kernel
//   64^2 plain fwd        45.8 us 844 TF/s   44.2 us  875 TF/s      64^2 fwd GN+SiLU            50.5 us  vs  52.3 us
//   64^2 plain dgrad      40.6 us 953 TF/s   39.9 us  968 TF/s      64^2 fwd +res+stats+save    63.5 us  vs  78.3 us
//   128^2 plain dgrad    147.3 us 1049 TF/s 140.1 us 1103 TF/s      64^2 dgrad + GN backward    55.0 us  vs  63.5 us
//   32^2 plain fwd        16.0 us            15.8 us                 32^2 fwd +res+stats+save    22.7 us  vs  25.4 us
// What it showed: removing the weight stream (the v2 kernel's L2 bound) is worth only 0-5 %, because the NEXT limit sits
// right behind it -- VALU ISSUE.  Per 128-pixel tile and SIMD the training-step forward launch needs ~8,600 cycles of VALU
// issue (GroupNorm+SiLU of the halo: 48 dword slices x 17 instructions, 4 of them transcendental; fp16 -> bf16 side
// output; bias / residual / rounding / statistics of 64 accumulator values per lane) beside 288 MFMAs = 9,216 cycles, of
// which each MFMA blocks the vector issue port for 8: 2,304 + 8,600 > 9,216, so even a perfectly interleaved stream is
// VALU-issue bound at ~85 % MFMA-busy, and a one-wave-per-SIMD kernel has no second wave to fill the unavoidable bubbles
// (dependent transcendental chains, waits).  The v2 kernel does the same VALU work with two waves per SIMD and ends up at
// the same place from the other side.  First version of this kernel (no pinned order): 1.3-2.3x SLOWER than v2 -- the
// compiler sank every B-fragment read next to its MFMAs (an LDS round trip per 2 MFMAs with nobody to hide it) and kept
// all 288 weight registers in VGPRs (spills); the fixes that got it to parity are the ones in the code: weights pinned in
// AGPRs through an empty asm constraint (the MFMA then reads its A operand from the accumulator file directly), the
// B reads / MFMAs / one dword slice of staging per k-block pinned with sched_group_barrier + sched_barrier, the slices
// anchored with an opaque asm (IR passes otherwise sink them to the LDS write), per-tile laundering of the per-thread
// address seeds (LICM otherwise hoists ~60 addresses into registers that do not exist).
// In-kernel stamps (diagnostic build -DWS_STAMPS + tools/ws_stamps.py; 128^2, 16 tiles per workgroup, cycles per tile of
// wave 0, medians; 9,216 = the 288 MFMAs alone):
//                 tile start   MFMA loop   barrier + epilogue   stores    whole tile
//   fwd plain        1,256       11,960          4,028           1,244      18,480
//   fwd GN+SiLU      1,292       17,056          4,012           1,228      23,572
//   fwd full         2,280       18,308          8,952           1,412      31,072
//   dgrad + GN bwd   2,864       11,852         10,212           1,296      26,240
// i.e. the pinned loop alone runs at 77 % MFMA-busy, the 816 staging VALU instructions ADD 5,100 cycles to it (6 cycles
// each: not hidden at all -- the wave issues in order, so every stall of the dependent v_exp -> v_add -> v_rcp -> v_mul
// chain of a slice also holds back the MFMA behind it), and the epilogue is 4-10 k cycles of exposed, dependency-bound
// VALU work.  A second wave per SIMD is what hides exactly these stalls in the v2 kernel.
// What would make it win: the slice chains modulo-scheduled over three k-blocks (exp of dword d, rcp of d-1, pack of d-2 per
// k-block: no dependent pair inside a k-block), fewer VALU instructions per tile (packed-fp16 SiLU halves the transform but costs forward
// accuracy: not taken), and the epilogue of tile t interleaved into the MFMAs of tile t+1 (needs a second accumulator
// set: 64 registers the weight-stationary layout does not have).  Not pursued further this round.
