// Single-head spatial self-attention of MONAI's SpatialAttentionBlock/SABlock (SURVEY.md §2.1 K6,
// Appendix A.1): softmax(q k^T * C^-0.5) v over L = H*W tokens with head dim = C, flash style —
// the L x L score matrix is never materialised (forward keeps one log-sum-exp per query for the
// backward, which recomputes the probabilities).  bf16 MFMA 32x32x16, fp32 softmax / accumulate.
//
// Data: qkv [B, L, 3C] bf16 (output of the fused to_q/to_k/to_v 1x1 conv), o / do [B, L, C] bf16.
// One workgroup = 4 waves = 128 tokens (32 per wave) of one image (two waves until late round 2: a streamed K/V tile then
// fed half as many MFMAs, and at head dim 256 the per-CU L2 -> LDS fill -- 76 KB per 32 keys -- outran the MFMAs 3:1;
// with four waves sharing a tile: dk/dv 867 -> 774 us, dq 438 -> 380, fwd 392 -> 356 at L = 4096, batch 8; head dim 128,
// L = 1024, batch 32: 181 -> 110, 99 -> 81, 60 -> 53 us); token tiles of 64 (32 for C=256)
// are staged in LDS as plain [token][d] rows with pitch 2*D+80 bytes (conflict-free ds_read_b128,
// near conflict-free ds_read_b64_tr_b16).
//
// MFMA orientation (see cdna_hip_programming.md §3 "accumulator tile as the next operand"):
//   S^T[key][query] = K Q^T           A = K rows,  B = Q rows          (plain 16-byte row reads)
//   O^T[d][query]  += V^T P^T         A = V^T via transposing read,   B = P^T straight from the
//                                      S^T accumulators (k order permuted: key 16s+8(j>>2)+4h+(j&3))
// the backward kernels use the same two patterns with the roles of q / k / v / do exchanged.
#include "pti_common.h"

namespace {

typedef short v4s __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));

constexpr int NW = 4;          // waves per workgroup
constexpr int TB = 32 * NW;    // tokens owned by a workgroup

template <int D>
struct ACfg {
  static constexpr int P = 2 * D + 80;          // LDS row pitch (bytes)
  static constexpr int KT = (D > 128) ? 32 : 64; // streamed token tile (backward kernels; the forward streams 64 at every head dim)
  static constexpr int NB = KT / 32;            // 32-token blocks per streamed tile
  static constexpr int KS = D / 16;             // MFMA k-steps over d
  static constexpr int DB = D / 32;             // 32-wide d blocks
};

// lane l reads the 8 consecutive d of token (t0 + (l&31)) at k-step ks: operand "token on M/N, d on K"
template <int D>
__device__ __forceinline__ bf16x8 frag_row(const unsigned char* tile, int t0, int ks, int lane) {
  return *(const bf16x8*)(tile + (t0 + (lane & 31)) * ACfg<D>::P + (16 * ks + 8 * (lane >> 5)) * 2);
}
// operand "d on M, token on K" in the k order of an accumulator-sourced partner operand:
// element j of lane half h <-> token t0 + 16*s + 8*(j>>2) + 4*h + (j&3); row = d0 + (l&31)
template <int D>
__device__ __forceinline__ bf16x8 frag_tr(const unsigned char* tile, int t0, int s, int d0, int lane) {
  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3, h = g >> 1;
  const unsigned char* p = tile + (t0 + 16 * s + 4 * h + q) * ACfg<D>::P + (d0 + 16 * (g & 1) + 4 * pp) * 2;
  const v4s a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(p));
  const v4s b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(p + 8 * ACfg<D>::P));
  v8s t = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, t);
}
// accumulator rows 8s..8s+7 -> bf16 operand fragment of k-step s
__device__ __forceinline__ bf16x8 acc_to_frag(const f32x16& x, int s) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (bf16)x[8 * s + j];
  return r;
}

// cooperative copy of `rows` token rows (D bf16 each) from global (row stride gstride elements) to LDS
// Streamed tiles go global -> registers -> LDS in two steps: all 16-byte pieces of a tile are ISSUED together
// (issue_rows) right after the previous tile was stored, so they fly under the MFMAs of the current tile, and are
// written to LDS (store_rows) at the top of the next iteration.  (A plain load/store loop waited for each piece in
// turn: 16 dependent L2 round trips per 64-token tile, ~6x the MFMA time at one wave per SIMD.)
template <int D, int NT, int ROWS>
struct RowRegs {
  static constexpr int N = ROWS * (D / 8) / NT;
  u32x4 r[N];
};
template <int D, int NT, int ROWS>
__device__ __forceinline__ void issue_rows(RowRegs<D, NT, ROWS>& rr, const bf16* src, size_t gstride, int valid, int tid) {
  constexpr int NC = D / 8;
#pragma unroll
  for (int i = 0; i < RowRegs<D, NT, ROWS>::N; ++i) {
    const int e = tid + i * NT, r = e / NC, c = e % NC;
    rr.r[i] = u32x4{0u, 0u, 0u, 0u};
    if (r < valid) rr.r[i] = *(const u32x4*)(src + (size_t)r * gstride + c * 8);
  }
}
template <int D, int NT, int ROWS>
__device__ __forceinline__ void store_rows(unsigned char* tile, const RowRegs<D, NT, ROWS>& rr, int tid) {
  constexpr int NC = D / 8;
#pragma unroll
  for (int i = 0; i < RowRegs<D, NT, ROWS>::N; ++i) {
    const int e = tid + i * NT, r = e / NC, c = e % NC;
    *(u32x4*)(tile + r * ACfg<D>::P + c * 16) = rr.r[i];
  }
}

// (rows at or beyond `valid` -- past the end of a token count that is not a multiple of the tile -- are zero-filled)
template <int D, int NT>
__device__ __forceinline__ void load_rows(unsigned char* tile, const bf16* src, size_t gstride, int rows, int valid, int tid) {
  constexpr int NC = D / 8;
  for (int e = tid; e < rows * NC; e += NT) {
    const int r = e / NC, c = e % NC;
    u32x4 v = u32x4{0u, 0u, 0u, 0u};
    if (r < valid) v = *(const u32x4*)(src + (size_t)r * gstride + c * 8);
    *(u32x4*)(tile + r * ACfg<D>::P + c * 16) = v;
  }
}

// acc[j] += A_j(ks) x B_j(ks) over the D / 16 k-steps of the head dimension, j < NJ independent accumulators: the A
// fragments (token rows of an LDS tile) are read TWO k-steps ahead into rotating registers and the issue order is pinned.
// As plain loops hipcc issued one ds_read_b128 into one register set right in front of every MFMA and waited for it
// (`s_waitcnt lgkmcnt(0)` before each of the 32 score MFMAs of a tile: ~150 exposed cycles per 32-cycle MFMA with one wave
// per SIMD), and with a single accumulator per loop the MFMAs were also a dependent chain.
template <int D, int NJ, class FA, class FB>
__device__ __forceinline__ void kloop(f32x16 (&acc)[NJ], FA&& fa, FB&& fb) {
  constexpr int KS = D / 16;
  bf16x8 buf[3][NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    buf[0][j] = fa(j, 0);
    buf[1][j] = fa(j, 1);
  }
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if (ks + 2 < KS) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) buf[(ks + 2) % 3][j] = fa(j, ks + 2);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(buf[ks % 3][j], fb(j, ks), acc[j], 0, 0, 0);
    if (ks + 2 < KS) {
      if constexpr (NJ == 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      else if constexpr (NJ == 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      else __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
    }
    if constexpr (NJ == 1) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    else if constexpr (NJ == 2) __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    else __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(64 * NW) void attn_fwd_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ o,
                                                           float* __restrict__ lse2, int L, float c_log2) {
  using C = ACfg<D>;
  constexpr int KTF = 64, NBF = KTF / 32;   // streamed key tile: 64 at every head dim (32 at D = 256 until the staging registers halved)
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * KTF * C::P];
  unsigned char* sK = smem;
  unsigned char* sV = smem + KTF * C::P;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y, q0 = blockIdx.x * TB + wave * 32;
  const size_t rs = 3 * D;  // row stride of qkv
  const bf16* base = qkv + (size_t)b * L * rs;

  bf16x8 bq[C::KS];
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks)
    bq[ks] = *(const bf16x8*)(base + (size_t)min(q0 + (lane & 31), L - 1) * rs + 16 * ks + 8 * (lane >> 5));

  f32x16 oacc[C::DB];
#pragma unroll
  for (int d = 0; d < C::DB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
  float m = -1e30f, lsum = 0.f;

  RowRegs<D, 64 * NW, KTF> rk, rv;
  issue_rows(rk, base + D, rs, L, tid);
  issue_rows(rv, base + 2 * D, rs, L, tid);
  for (int k0 = 0; k0 < L; k0 += KTF) {
    __syncthreads();
    store_rows<D, 64 * NW, KTF>(sK, rk, tid);
    store_rows<D, 64 * NW, KTF>(sV, rv, tid);
    __syncthreads();
    if (k0 + KTF < L) {
      issue_rows(rk, base + (size_t)(k0 + KTF) * rs + D, rs, L - k0 - KTF, tid);
      issue_rows(rv, base + (size_t)(k0 + KTF) * rs + 2 * D, rs, L - k0 - KTF, tid);
    }
    f32x16 sacc[NBF];
#pragma unroll
    for (int kb = 0; kb < NBF; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
    kloop<D, NBF>(sacc, [&](int kb, int ks) { return frag_row<D>(sK, kb * 32, ks, lane); }, [&](int, int ks) { return bq[ks]; });
    // online softmax over the keys of this tile (query = lane & 31; the two lane halves hold
    // different key rows of the same query)
    float mt = -1e30f;
#pragma unroll
    for (int kb = 0; kb < NBF; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        sacc[kb][r] *= c_log2;
        // keys past the end of a ragged sequence (last tile only) take no probability mass
        if (k0 + KTF > L && k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) >= L) sacc[kb][r] = -1e30f;
        mt = fmaxf(mt, sacc[kb][r]);
      }
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float mn = fmaxf(m, mt);
    const float alpha = exp2f(m - mn);
    m = mn;
    float ps = 0.f;
#pragma unroll
    for (int kb = 0; kb < NBF; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = exp2f(sacc[kb][r] - mn);
        sacc[kb][r] = p;
        ps += p;
      }
    lsum = lsum * alpha + ps;
    // rescale the output accumulators only when some query of this wave saw a new maximum in this tile (alpha == 1
    // exactly otherwise: same bits).  After the first few key tiles that is rare, and the rescale -- D/2 multiplies per
    // lane, through AGPR copies at head dim 256 -- costs more issue slots than the tile's MFMAs.
    if (__any(alpha != 1.0f)) {
#pragma unroll
      for (int d = 0; d < C::DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
    }
#pragma unroll
    for (int kb = 0; kb < NBF; ++kb)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 bp = acc_to_frag(sacc[kb], s);
#pragma unroll
        for (int d = 0; d < C::DB; ++d)
          oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(sV, kb * 32, s, d * 32, lane), bp, oacc[d], 0, 0, 0);
      }
  }
  lsum += __shfl_xor(lsum, 32, 64);
  const float inv = 1.f / lsum;
  const int qi = q0 + (lane & 31), h = lane >> 5;
  if (qi >= L) return;
  bf16* orow = o + ((size_t)b * L + qi) * D;
#pragma unroll
  for (int d = 0; d < C::DB; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *(u32x2*)(orow + d * 32 + 8 * g + 4 * h) =
          pack4(oacc[d][4 * g] * inv, oacc[d][4 * g + 1] * inv, oacc[d][4 * g + 2] * inv, oacc[d][4 * g + 3] * inv);
  if (h == 0) lse2[(size_t)b * L + qi] = m + log2f(lsum);
}

// delta[b][q] = sum_d do*o
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16* __restrict__ o, const bf16* __restrict__ dout,
                                                         float* __restrict__ delta, long long rows, int D) {
  const int NC = D / 8, rpb = 256 / NC;
  const int lc = threadIdx.x % NC, lr = threadIdx.x / NC;
  for (long long r0 = (long long)blockIdx.x * rpb; r0 < rows; r0 += (long long)gridDim.x * rpb) {
    const long long r = r0 + lr;
    float s = 0.f;
    if (r < rows) {
      float a[8], b[8];
      unpack8(*(const u32x4*)(o + r * D + lc * 8), a);
      unpack8(*(const u32x4*)(dout + r * D + lc * 8), b);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += a[j] * b[j];
    }
    for (int ofs = NC >> 1; ofs > 0; ofs >>= 1) s += __shfl_xor(s, ofs, 64);
    if (r < rows && lc == 0) delta[r] = s;
  }
}

// ------------------------------------------------------------------------------------------------
// dQ: workgroup owns 64 queries, streams key tiles.  dS^T[key][query] = P^T (dP^T - delta) * scale,
// dQ^T[d][query] += K^T dS^T.
template <int D>
__global__ __launch_bounds__(64 * NW) void attn_bwd_dq_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ dout,
                                                              const float* __restrict__ lse2,
                                                              const float* __restrict__ delta, bf16* __restrict__ dqkv,
                                                              int L, float c_log2, float scale) {
  using C = ACfg<D>;
  constexpr int KTF = C::KT, NBF = C::NB;   // (64-token tiles at head dim 256 measured 5 % slower here, 14 % faster in the forward)
  // Q / dO fragments live in registers: LDS only holds the streamed K/V tile.  (Head dim 256 used to keep them in LDS:
  // 113 KB per 2-wave workgroup = ONE workgroup per CU, half the SIMDs idle; in registers -- 448 of the 512 per lane --
  // it is 38 KB and two workgroups per CU: 838 -> 413 us at L=4096, batch 8.)
  constexpr bool REGQ = true;
  __shared__ __attribute__((aligned(16))) unsigned char smem[((REGQ ? 0 : 2 * TB) + 2 * KTF) * C::P];
  unsigned char* sQ = smem;
  unsigned char* sDO = sQ + (REGQ ? 0 : TB) * C::P;
  unsigned char* sK = sDO + (REGQ ? 0 : TB) * C::P;
  unsigned char* sV = sK + KTF * C::P;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y, qb0 = blockIdx.x * TB;
  const size_t rs = 3 * D;
  const bf16* base = qkv + (size_t)b * L * rs;
  const int qi = qb0 + wave * 32 + (lane & 31);
  const int qc = min(qi, L - 1);   // ragged tail: compute on a clamped row, store nothing
  bf16x8 bq[REGQ ? C::KS : 1], bdo[REGQ ? C::KS : 1];
  if constexpr (REGQ) {
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      bq[ks] = *(const bf16x8*)(base + (size_t)qc * rs + 16 * ks + 8 * (lane >> 5));
      bdo[ks] = *(const bf16x8*)(dout + ((size_t)b * L + qc) * D + 16 * ks + 8 * (lane >> 5));
    }
  } else {
    load_rows<D, 64 * NW>(sQ, base + (size_t)qb0 * rs, rs, TB, L - qb0, tid);
    load_rows<D, 64 * NW>(sDO, dout + ((size_t)b * L + qb0) * D, D, TB, L - qb0, tid);
  }
  const float my_lse = lse2[(size_t)b * L + qc], my_delta = delta[(size_t)b * L + qc];
  f32x16 dq[C::DB];
#pragma unroll
  for (int d = 0; d < C::DB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[d][r] = 0.f;

  RowRegs<D, 64 * NW, KTF> rk, rv;
  issue_rows(rk, base + D, rs, L, tid);
  issue_rows(rv, base + 2 * D, rs, L, tid);
  for (int k0 = 0; k0 < L; k0 += KTF) {
    __syncthreads();
    store_rows<D, 64 * NW, KTF>(sK, rk, tid);
    store_rows<D, 64 * NW, KTF>(sV, rv, tid);
    __syncthreads();
    if (k0 + KTF < L) {
      issue_rows(rk, base + (size_t)(k0 + KTF) * rs + D, rs, L - k0 - KTF, tid);
      issue_rows(rv, base + (size_t)(k0 + KTF) * rs + 2 * D, rs, L - k0 - KTF, tid);
    }
#pragma unroll
    for (int kb = 0; kb < NBF; ++kb) {
      f32x16 sa, dp;
      if constexpr (REGQ && D <= 128) {     // K / V fragment reads two k-steps ahead (see kloop; head dim 256 has no registers left)
        f32x16 sd[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) sd[0][r] = sd[1][r] = 0.f;
        kloop<D, 2>(sd, [&](int j, int ks) { return frag_row<D>(j ? sV : sK, kb * 32, ks, lane); },
                    [&](int j, int ks) { return j ? bdo[ks] : bq[ks]; });
        sa = sd[0];
        dp = sd[1];
      } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) sa[r] = dp[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        bf16x8 fq, fdo;
        if constexpr (REGQ) { fq = bq[ks]; fdo = bdo[ks]; }
        else { fq = frag_row<D>(sQ, wave * 32, ks, lane); fdo = frag_row<D>(sDO, wave * 32, ks, lane); }
        sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row<D>(sK, kb * 32, ks, lane), fq, sa, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row<D>(sV, kb * 32, ks, lane), fdo, dp, 0, 0, 0);
      }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float p = exp2f(sa[r] * c_log2 - my_lse);
        if (k0 + KTF > L && k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) >= L) p = 0.f;   // ragged tail keys
        sa[r] = p * (dp[r] - my_delta) * scale;
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 bds = acc_to_frag(sa, s);
#pragma unroll
        for (int d = 0; d < C::DB; ++d)
          dq[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(sK, kb * 32, s, d * 32, lane), bds, dq[d], 0, 0, 0);
      }
    }
  }
  const int h = lane >> 5;
  if (qi >= L) return;
  bf16* orow = dqkv + ((size_t)b * L + qi) * rs;
#pragma unroll
  for (int d = 0; d < C::DB; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *(u32x2*)(orow + d * 32 + 8 * g + 4 * h) = pack4(dq[d][4 * g], dq[d][4 * g + 1], dq[d][4 * g + 2], dq[d][4 * g + 3]);
}

// ------------------------------------------------------------------------------------------------
// dK, dV: workgroup owns 64 keys, streams query tiles.  P[query][key] (key on the lane):
// dV^T[d][key] += dO^T P,  dK^T[d][key] += Q^T dS.  DSPLIT splits d over blockIdx.z to bound VGPRs (no longer used: with
// four waves per workgroup the streamed-tile staging registers halved and head dim 256 fits unsplit -- 512 registers and
// 40 spilled -- which drops the second computation of S and dP and the second pass over Q/dO: 774 -> 529 us).
template <int D, int DSPLIT>
__global__ __launch_bounds__(64 * NW) void attn_bwd_dkdv_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ dout,
                                                                const float* __restrict__ lse2,
                                                                const float* __restrict__ delta, bf16* __restrict__ dqkv,
                                                                int L, float c_log2, float scale) {
  using C = ACfg<D>;
  constexpr int DBL = C::DB / DSPLIT;  // d blocks accumulated by this workgroup
  constexpr bool REGK = true;    // K / V fragments of the owned keys live in registers (head dim 256 too: 1542 -> 815 us, see dq)
  __shared__ __attribute__((aligned(16))) unsigned char smem[((REGK ? 0 : 2 * TB) + 2 * C::KT) * C::P + 2 * C::KT * 4];
  unsigned char* sK = smem;
  unsigned char* sV = sK + (REGK ? 0 : TB) * C::P;
  unsigned char* sQ = sV + (REGK ? 0 : TB) * C::P;
  unsigned char* sDO = sQ + C::KT * C::P;
  float* sL = reinterpret_cast<float*>(sDO + C::KT * C::P);
  float* sDl = sL + C::KT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y, kb0 = blockIdx.x * TB, dz = blockIdx.z * DBL;
  const size_t rs = 3 * D;
  const bf16* base = qkv + (size_t)b * L * rs;
  bf16x8 bk[REGK ? C::KS : 1], bv[REGK ? C::KS : 1];
  if constexpr (REGK) {
    const bf16* krow = base + (size_t)min(kb0 + wave * 32 + (lane & 31), L - 1) * rs + 8 * (lane >> 5);
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      bk[ks] = *(const bf16x8*)(krow + D + 16 * ks);
      bv[ks] = *(const bf16x8*)(krow + 2 * D + 16 * ks);
    }
  } else {
    load_rows<D, 64 * NW>(sK, base + (size_t)kb0 * rs + D, rs, TB, L - kb0, tid);
    load_rows<D, 64 * NW>(sV, base + (size_t)kb0 * rs + 2 * D, rs, TB, L - kb0, tid);
  }
  f32x16 dk[DBL], dv[DBL];
#pragma unroll
  for (int d = 0; d < DBL; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) dk[d][r] = dv[d][r] = 0.f;
  const int h = lane >> 5;

  RowRegs<D, 64 * NW, C::KT> rq, rdo;
  issue_rows(rq, base, rs, L, tid);
  issue_rows(rdo, dout + (size_t)b * L * D, D, L, tid);
  for (int q0 = 0; q0 < L; q0 += C::KT) {
    __syncthreads();
    store_rows<D, 64 * NW, C::KT>(sQ, rq, tid);
    store_rows<D, 64 * NW, C::KT>(sDO, rdo, tid);
    for (int e = tid; e < C::KT; e += 64 * NW) {
      // queries past a ragged end: log-sum-exp = +huge => probability exactly 0 below
      const bool ok = q0 + e < L;
      sL[e] = ok ? lse2[(size_t)b * L + q0 + e] : 1e30f;
      sDl[e] = ok ? delta[(size_t)b * L + q0 + e] : 0.f;
    }
    __syncthreads();
    if (q0 + C::KT < L) {
      issue_rows(rq, base + (size_t)(q0 + C::KT) * rs, rs, L - q0 - C::KT, tid);
      issue_rows(rdo, dout + ((size_t)b * L + q0 + C::KT) * D, D, L - q0 - C::KT, tid);
    }
#pragma unroll
    for (int qb = 0; qb < C::NB; ++qb) {
      f32x16 sa, dp;
      if constexpr (REGK && D <= 128) {     // Q / dO fragment reads two k-steps ahead (see kloop)
        f32x16 sd[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) sd[0][r] = sd[1][r] = 0.f;
        kloop<D, 2>(sd, [&](int j, int ks) { return frag_row<D>(j ? sDO : sQ, qb * 32, ks, lane); },
                    [&](int j, int ks) { return j ? bv[ks] : bk[ks]; });
        sa = sd[0];
        dp = sd[1];
      } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) sa[r] = dp[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        bf16x8 fk, fv;
        if constexpr (REGK) { fk = bk[ks]; fv = bv[ks]; }
        else { fk = frag_row<D>(sK, wave * 32, ks, lane); fv = frag_row<D>(sV, wave * 32, ks, lane); }
        sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row<D>(sQ, qb * 32, ks, lane), fk, sa, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row<D>(sDO, qb * 32, ks, lane), fv, dp, 0, 0, 0);
      }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qrow = qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;  // accumulator row -> query of the tile
        const float p = exp2f(sa[r] * c_log2 - sL[qrow]);
        dp[r] = p * (dp[r] - sDl[qrow]) * scale;  // dS
        sa[r] = p;                                // P
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 bp = acc_to_frag(sa, s), bds = acc_to_frag(dp, s);
#pragma unroll
        for (int d = 0; d < DBL; ++d) {
          dv[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(sDO, qb * 32, s, (dz + d) * 32, lane), bp, dv[d], 0, 0, 0);
          dk[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(sQ, qb * 32, s, (dz + d) * 32, lane), bds, dk[d], 0, 0, 0);
        }
      }
    }
  }
  const int ki = kb0 + wave * 32 + (lane & 31);
  if (ki >= L) return;
  bf16* orow = dqkv + ((size_t)b * L + ki) * rs;
#pragma unroll
  for (int d = 0; d < DBL; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int dd = (dz + d) * 32 + 8 * g + 4 * h;
      *(u32x2*)(orow + D + dd) = pack4(dk[d][4 * g], dk[d][4 * g + 1], dk[d][4 * g + 2], dk[d][4 * g + 3]);
      *(u32x2*)(orow + 2 * D + dd) = pack4(dv[d][4 * g], dv[d][4 * g + 1], dv[d][4 * g + 2], dv[d][4 * g + 3]);
    }
}

}  // namespace

static int attn_check(const char* who, int b, int l, int c) {
  if (b <= 0 || l <= 0) PTI_FAIL(PTI_EINVAL, "%s: bad dims", who);
  if (c != 64 && c != 128 && c != 256) PTI_FAIL(PTI_EUNSUPPORTED, "%s: head dim %d (supported 64, 128, 256)", who, c);
  return 0;
}

extern "C" int pti_attention_fwd(const void* qkv, void* o, float* lse2, int b, int l, int c, pti_stream_t s) {
  if (!qkv || !o || !lse2) PTI_FAIL(PTI_EINVAL, "attention_fwd: null pointer");
  if (int rc = attn_check("attention_fwd", b, l, c)) return rc;
  const float c_log2 = 1.4426950408889634f / sqrtf((float)c);
  dim3 grid((l + TB - 1) / TB, b), blk(64 * NW);
  hipStream_t st = (hipStream_t)s;
  if (c == 64) PTI_LAUNCH(attn_fwd_kernel<64>, grid, blk, 0, st, (const bf16*)qkv, (bf16*)o, lse2, l, c_log2);
  else if (c == 128) PTI_LAUNCH(attn_fwd_kernel<128>, grid, blk, 0, st, (const bf16*)qkv, (bf16*)o, lse2, l, c_log2);
  else PTI_LAUNCH(attn_fwd_kernel<256>, grid, blk, 0, st, (const bf16*)qkv, (bf16*)o, lse2, l, c_log2);
  PTI_CHECK_LAUNCH("attention_fwd");
  return PTI_OK;
}

extern "C" int pti_attention_bwd(const void* qkv, const void* o, const void* dout, const float* lse2, float* delta,
                                 void* dqkv, int b, int l, int c, pti_stream_t s) {
  if (!qkv || !o || !dout || !lse2 || !delta || !dqkv) PTI_FAIL(PTI_EINVAL, "attention_bwd: null pointer");
  if (int rc = attn_check("attention_bwd", b, l, c)) return rc;
  const float scale = 1.0f / sqrtf((float)c), c_log2 = 1.4426950408889634f * scale;
  hipStream_t st = (hipStream_t)s;
  const long long rows = (long long)b * l;
  long long db = (rows + (256 / (c / 8)) - 1) / (256 / (c / 8));
  if (db > 4096) db = 4096;
  PTI_LAUNCH(attn_delta_kernel, dim3((unsigned)db), dim3(256), 0, st, (const bf16*)o, (const bf16*)dout, delta, rows, c);
  PTI_CHECK_LAUNCH("attention_delta");
  dim3 grid((l + TB - 1) / TB, b), blk(64 * NW);
  const bf16* Q = (const bf16*)qkv; const bf16* DO = (const bf16*)dout; bf16* DQ = (bf16*)dqkv;
  if (c == 64) {
    PTI_LAUNCH(attn_bwd_dq_kernel<64>, grid, blk, 0, st, Q, DO, lse2, delta, DQ, l, c_log2, scale);
    PTI_LAUNCH((attn_bwd_dkdv_kernel<64, 1>), grid, blk, 0, st, Q, DO, lse2, delta, DQ, l, c_log2, scale);
  } else if (c == 128) {
    PTI_LAUNCH(attn_bwd_dq_kernel<128>, grid, blk, 0, st, Q, DO, lse2, delta, DQ, l, c_log2, scale);
    PTI_LAUNCH((attn_bwd_dkdv_kernel<128, 1>), grid, blk, 0, st, Q, DO, lse2, delta, DQ, l, c_log2, scale);
  } else {
    PTI_LAUNCH(attn_bwd_dq_kernel<256>, grid, blk, 0, st, Q, DO, lse2, delta, DQ, l, c_log2, scale);
    PTI_LAUNCH((attn_bwd_dkdv_kernel<256, 1>), dim3((l + TB - 1) / TB, b, 1), blk, 0, st, Q, DO, lse2, delta, DQ, l, c_log2, scale);
  }
  PTI_CHECK_LAUNCH("attention_bwd");
  return PTI_OK;
}
