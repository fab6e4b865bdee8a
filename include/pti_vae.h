/*
 * pti_vae.h — C-ABI of libpti_vae_hip.so: hand-written HIP (gfx950 / CDNA4) kernels for the
 * VAE training hot path of Sukikui/PTI-LDM-VAE.
 *
 * The reference has no FFI: its only seam is the Python class pti_ldm_vae.models.VAEModel
 * (reference src/pti_ldm_vae/models/autoencoder.py:6-171), whose arithmetic is MONAI's
 * AutoencoderKL reached through torch.nn ops (autoencoder.py:3,67-79,114).  Each entry point
 * below replaces one of those implicit ATen/cuDNN kernels; the comment on each names the
 * reference call it stands in for.  Host side: pti_ldm_vae_amd/_lib.py binds them with ctypes.
 *
 * Conventions (all entry points):
 *   - plain device pointers + sizes, no torch types; launches ONLY on the given stream;
 *   - never allocates, never synchronises, never throws;
 *   - returns 0 on success or a negative PTI_E* code; pti_last_error_string() explains it;
 *   - activations are NHWC bf16 (channels innermost) unless a parameter says otherwise;
 *   - "stats" buffers are int64_t[N][G][2] = Q47.16 fixed-point {sum, sum of squares} over one
 *     (sample, group) -- see the note above pti_conv_desc; consumers turn them into mean / rstd
 *     themselves (count and eps are passed along).
 */
#ifndef PTI_VAE_H
#define PTI_VAE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* pti_stream_t; /* hipStream_t */

#define PTI_ABI_VERSION 5   /* 5 (round 3): + pti_direct_repack, pti_pad_nchw_to_nhwc32 / pti_slice_nhwc32_to_nchw,
                                 pti_conv2d_mfma_gnbwd_chain (+ _supported), pti_gn_affine_grads, pti_gn_sums_finalize_affine */

#define PTI_OK 0
#define PTI_EINVAL (-1)   /* bad pointer / dimension */
#define PTI_EUNSUPPORTED (-2) /* channel multiple / mode not built */
#define PTI_ELAUNCH (-3)  /* hipGetLastError() after launch */

/* input gather modes of the implicit-GEMM convolution */
#define PTI_CONV_S1 0   /* stride 1, pad (k-1)/2              nn.Conv2d(k, s=1, p=k//2)              */
#define PTI_CONV_S2PAD 1 /* F.pad(x,(0,1,0,1)) + 3x3 stride 2   MONAI AEKLDownsample                   */
#define PTI_CONV_UP2 2  /* nearest 2x upsample then 3x3 s1 p1  MONAI Upsample(nontrainable)+postconv  */
#define PTI_CONV_ZINS 3 /* zero-insert 2x, pad_lo 2: data-gradient of PTI_CONV_S2PAD                  */

#define PTI_PRO_NONE 0
#define PTI_PRO_GN 1      /* x -> GroupNorm affine                nn.GroupNorm                          */
#define PTI_PRO_GN_SILU 2 /* x -> GroupNorm affine -> SiLU        nn.GroupNorm + F.silu (AEKLResBlock)  */

/* GroupNorm statistics travel as int64 Q47.16 fixed-point {sum, sum of squares} per (sample, group) --           *
 * `int64_t stats[n][groups][2]`, value = integer / 65536 -- so that accumulating them with atomics is order-       *
 * independent and every result downstream is bitwise reproducible (see pti_common.h).                              */
typedef struct pti_conv_desc {
  int32_t n, h, w, cin;   /* real input tensor [n,h,w,cin]                                   */
  int32_t ho, wo, cout;   /* output tensor [n,ho,wo,cout]                                    */
  int32_t ksize;          /* 1 or 3                                                          */
  int32_t mode;           /* PTI_CONV_*                                                      */
  int32_t prologue;       /* PTI_PRO_*: applied to the input while it is staged into LDS     */
  int32_t groups;         /* GroupNorm groups of the prologue                                */
  int32_t add_residual;   /* epilogue: y += residual (same shape as y)                       */
  int32_t accum_stats;    /* epilogue: atomically add {sum,sumsq} of the stored y per        */
                          /* (sample, out-group) into out_stats; out_groups below            */
  int32_t out_groups;
  float eps;              /* GroupNorm eps of the prologue                                   */
  int32_t in_f32;         /* direct conv only: input is fp32 with explicit element strides   */
  int32_t out_f32;        /* direct conv only: output is fp32 with explicit element strides  */
  int64_t in_stride[4];   /* n,h,w,c element strides when in_f32 (else NHWC dense)           */
  int64_t out_stride[4];  /* n,h,w,c element strides when out_f32                            */
  /* 16-bit storage format of the NHWC activation operands: 0 = bf16, 1 = IEEE fp16.  The forward */
  /* pass may keep its activations in fp16 (same bytes, 8x finer rounding); MFMA operands, saved  */
  /* activated inputs, attention tensors and all gradients are bf16.  res_f16 also describes the  */
  /* GroupNorm input `gx` of pti_conv2d_mfma_gnbwd.                                               */
  int32_t in_f16, res_f16, out_f16;
  /* pti_conv2d_mfma only: y is [n][ho/2][wo/2][cout] = the 2x2 SUM pool of the conv output (the data      */
  /* gradient of nn.Upsample(nearest, 2x) + conv, fused: the full-resolution gradient is never written).   */
  int32_t pool2x2_out;
  /* pti_conv2d_mfma / _saveact only: w_packed was packed with w_f16 = 1 and the MFMA multiplies fp16 operands      */
  /* (v_mfma_f32_32x32x16_f16; same rate as bf16, 8x finer operand rounding).  Needs in_f16 = out_f16 = 1 (and      */
  /* res_f16 when a residual is added): the forward convs on fp16 storage.  Gradients always use bf16 operands.     */
  int32_t w_f16;
  /* pti_conv2d_mfma only: y = max(conv + bias [+ residual], 0) -- the ReLU of the perceptual network's Fire modules  */
  /* fused into the store.  Plain fp16 forward launches (w_f16, no prologue, stride-1 gather); refused elsewhere.     */
  int32_t relu_out;
} pti_conv_desc;

int pti_abi_version(void);
const char* pti_last_error_string(void);
/* Symbol of the (last) kernel the calling thread's most recent pti_* call launched, demangled, as the HIP runtime and
 * rocprofv3 name it; "" before the first launch.  Diagnostics: lets a caller pair its own timings with profiler rows. */
const char* pti_last_kernel_name(void);

/* ---- weights ------------------------------------------------------------------------- */
/* Bytes of the MFMA-packed bf16 image of a [cout,cin,k,k] fp32 weight (0 if unsupported).  */
int64_t pti_conv_packed_bytes(int cout, int cin, int ksize, int mode);
/* Pack fp32 OIHW master weights (nn.Conv2d.weight / nn.Linear.weight layout) into the MFMA
 * fragment order read by pti_conv2d_mfma.  transpose_flip=1 builds the data-gradient operand
 * W'[ci][co][2-kh][2-kw]; cout/cin are those of the ORIGINAL weight.  nsrc>1 concatenates
 * nsrc weights along the output channels (to_q/to_k/to_v fused into one 1x1).             */
int pti_conv_pack_weights(const float* const* w_oihw, int nsrc, void* packed, int cout, int cin,
                          int ksize, int mode, int transpose_flip, int w_f16, pti_stream_t s);

/* Batched form (one launch for all layers of the model after an optimiser step): the caller keeps a
 * table of pti_conv_pack_entry_bytes()-byte entries; pti_conv_pack_table_fill writes ONE entry into HOST
 * memory and returns its size in 256-element blocks; the caller uploads the table plus the running
 * first-block index of every entry, then launches everything with pti_conv_pack_weights_batched.   */
int pti_conv_pack_entry_bytes(void);
int pti_conv_pack_table_fill(void* host_entry, const float* w_oihw_dev, void* packed_dev, int cout,
                             int cin, int ksize, int mode, int transpose_flip, int w_f16, int64_t* nblocks);
int pti_conv_pack_weights_batched(const void* table_dev, const int* blk_first_dev, int n,
                                  int total_blocks, pti_stream_t s);

/* ---- GroupNorm statistics (nn.GroupNorm's reduction) ---------------------------------- */
/* stats[n][g] += {sum, sumsq} of x[n, :, channels of g]; stats must be zeroed by the caller. */
int pti_gn_stats(const void* x_nhwc_16bit, int64_t* stats, int n, int hw, int c, int groups,
                 int x_f16, pti_stream_t s);

/* ---- convolutions ---------------------------------------------------------------------- */
/* Implicit-GEMM 3x3 / 1x1 convolution on bf16 MFMA (v_mfma_f32_32x32x16_bf16), fp32 accumulate:
 * y = conv(prologue(x)) + bias [+ residual].  Replaces nn.Conv2d (+ the GroupNorm/SiLU in
 * front of it, + the residual add behind it) inside MONAI AEKLResBlock / AEKLDownsample /
 * Upsample / SABlock linears.  cin, cout multiples of 32.  in_stats / out_stats: int64_t[n][groups][2]
 * Q47.16 fixed-point sums (above).                                                          */
int pti_conv2d_mfma(const void* x, const void* w_packed, const float* bias, const int64_t* in_stats,
                    const float* gamma, const float* beta, const void* residual, void* y,
                    int64_t* out_stats, const pti_conv_desc* d, pti_stream_t s);

/* Same launch, plus a side output: act_out = prologue(x) as bf16 NHWC [n][h][w][cin] (the tensor autograd
 * would save for nn.Conv2d's weight gradient).  3x3 PTI_CONV_S1 with a GroupNorm(+SiLU) prologue only.  The
 * weight-gradient call then reads act_out with PTI_PRO_NONE instead of re-applying the prologue. */
int pti_conv2d_mfma_saveact(const void* x, const void* w_packed, const float* bias, const int64_t* in_stats,
                            const float* gamma, const float* beta, const void* residual, void* y,
                            int64_t* out_stats, void* act_out, const pti_conv_desc* d, pti_stream_t s);

/* Direct (VALU, fp32 math) convolution for the degenerate-channel layers (cin or cout < 32):
 * conv_in, conv_out of Encoder/Decoder.  w: fp32 [k*k][cin][cout]; see pti_conv_desc strides. */
int pti_conv2d_direct(const void* x, const float* w_tck, const float* bias, const int64_t* in_stats,
                      const float* gamma, const float* beta, void* y, const pti_conv_desc* d,
                      pti_stream_t s);

/* Derived operands of the degenerate-channel convs (conv_in / conv_out of MONAI's Encoder / Decoder) in one launch:
 * per entry, from the fp32 master weight w [cout][cin][3][3]:  w_tck [9][cin][cout] (forward operand of pti_conv2d_direct),
 * w_tck_t [9][cout][cin] with the taps reversed (its data-gradient operand), wpad = the weight copied into a zero-padded
 * master [cout'][pad_cin][3][3] (rows co < cout, columns ci < cin; the rest of the buffer is left as it is -- allocate it
 * zeroed) that the MFMA packer reads, and bpad[0..cout) = b.  Any of the outputs may be NULL. */
#define PTI_DIRECT_REPACK_MAX 8
typedef struct {
  const float* w; const float* b;
  float* w_tck; float* w_tck_t; float* wpad; float* bpad;
  int cout, cin, pad_cin, reserved;
} pti_direct_repack_entry;
typedef struct {
  pti_direct_repack_entry e[PTI_DIRECT_REPACK_MAX];
  int n;
} pti_direct_repack_table;
int pti_direct_repack(const pti_direct_repack_table* t, pti_stream_t s);

/* Weight/bias gradient of pti_conv2d_direct (autograd of nn.Conv2d for the degenerate layers):
 * dw[tap*st_tap + cw*st_cw + k*st_k] += sum_p narrow[p][k] * P(wide)[p + sgn*(tap offset)][cw]
 * for every narrow channel k < cn; wide is dense NHWC bf16 with cw channels (prologue P optional),
 * narrow is fp32/bf16 with element strides narrow_stride[n,h,w,c].  dbias_wide[cw] += column sums
 * of wide, dbias_narrow[k] += sum of narrow (either may be NULL).  Per-block partials go to `workspace`
 * (plain stores) and are summed in block order by a second launch: deterministic, += into the outputs.  */
int pti_wgrad_direct(const void* wide, const void* narrow, float* dw, float* dbias_wide,
                     float* dbias_narrow, const int64_t* in_stats, const float* gamma,
                     const float* beta, int n, int h, int w, int cw, int cn, int ksize, int sgn,
                     int prologue, int groups, float eps, int narrow_f32, int wide_f16,
                     const int64_t* narrow_stride, int64_t dw_stride_tap, int64_t dw_stride_cw,
                     int64_t dw_stride_k, void* workspace, int64_t workspace_bytes, pti_stream_t s);

/* ---- convolution weight gradient (autograd of nn.Conv2d, MFMA path) ----------------------- */
int64_t pti_conv_wgrad_workspace_bytes(int cout, int cin, int ksize, int splits);
/* dw[cout][cin][k][k] (fp32, OIHW) and dbias[cout] (=, or += when accumulate) from dy and the
 * SAME x / prologue / mode the forward conv saw (prologue is recomputed in the loader).  Split-K
 * partials go to `workspace` with plain stores and are summed in a fixed order (deterministic). */
int pti_conv_wgrad_mfma(const void* x, const void* dy, const int64_t* in_stats, const float* gamma,
                        const float* beta, float* dw, float* dbias, void* workspace,
                        int64_t workspace_bytes, int accumulate, const pti_conv_desc* d,
                        pti_stream_t s);
/* The same work as two calls (pti_conv_wgrad_mfma is exactly partials + reduce): the split-K partial launch,
 * which reports in *splits_out an opaque slab token (slab count, plus a flag bit for the slab layout the kernel it
 * picked writes), and the fixed-order slab reduction into dw / dbias, which takes that token as `splits`.  Lets a
 * caller time or overlap the two launches separately. */
int pti_conv_wgrad_mfma_partials(const void* x, const void* dy, const int64_t* in_stats,
                                 const float* gamma, const float* beta, void* workspace,
                                 int64_t workspace_bytes, const pti_conv_desc* d, int* splits_out,
                                 pti_stream_t s);
int pti_conv_wgrad_reduce(const void* workspace, int splits, float* dw, float* dbias,
                          int accumulate, const pti_conv_desc* d, pti_stream_t s);

/* Batched form for the training step: up to PTI_WGRAD_BATCH_MAX independent weight-gradient problems of plain
 * stride-1 3x3 convs on bf16 inputs WITHOUT prologue (x = the saved activated input) in one partial launch + one
 * reduction launch.  A launch has ~11 us of fixed cost (dispatch, ring fill, cross-wave reduction, slab drain) against
 * 10..45 us of streaming per layer, and the layers' weight gradients are independent of each other, so the engine
 * collects them while backward walks the layers and flushes a batch at a time.  Deterministic like the single form.  */
#define PTI_WGRAD_BATCH_MAX 16
typedef struct pti_wgrad_job {
  const void* x;    /* bf16 [n,h,w,cin] */
  const void* dy;   /* bf16 [n,h,w,cout] */
  float* dw;        /* fp32 [cout,cin,3,3] */
  float* dbias;     /* fp32 [cout] or NULL */
  int32_t n, h, w, cin, cout;
  int32_t accumulate;   /* += instead of = */
} pti_wgrad_job;
int pti_conv_wgrad_mfma_batched(const pti_wgrad_job* jobs, int njobs, void* workspace, int64_t workspace_bytes,
                                pti_stream_t s);

/* ---- GroupNorm(+SiLU) backward, 2x2 sum pool ---------------------------------------------- */
/* dx = d/dx of act(GroupNorm(x)) given da (+ dres added), dgamma/dbeta += ; sums: float [n][c][2] scratch
 * (written, need not be zeroed); partials: float scratch of n * pti_gn_bwd_blocks(n, hw, c) * c * 2 elements;
 * stats as produced by pti_gn_stats on x.  (autograd of nn.GroupNorm+F.silu)
 * No floating-point atomics anywhere on this path: every workgroup stores its partial {sum dy, sum dy*xhat} row,
 * pti_gn_sums_finalize's kernel adds the rows in a fixed order and ONE workgroup of the apply kernel folds the
 * samples into dgamma / dbeta, so the backward pass is bitwise reproducible run to run.          */
int pti_gn_bwd_blocks(int n, int hw, int c);
int pti_gn_bwd(const void* x, const void* da, const void* dres, void* dx, const int64_t* stats,
               const float* gamma, const float* beta, float* sums, float* partials, float* dgamma,
               float* dbeta, int n, int hw, int c, int groups, float eps, int silu, int x_f16,
               pti_stream_t s);
/* Fused form used by the engine: the data-gradient conv computes dy = dA * act'(GN(gx)) in its epilogue and
 * stores, per pixel tile, the partial row gpartials[n][tile][c] = {sum dy, sum dy*xhat} (gx = the GroupNorm input,
 * same shape as the conv output; d->groups / d->eps describe that GroupNorm; d is a plain stride-1 / zero-insert
 * launch, w_packed the transposed+flipped pack; gpartials holds n * pti_conv_gnbwd_tiles(d) * cout * 2 floats).
 * pti_gn_sums_finalize(gpartials, sums, n, tiles, 2*cout) adds the tile rows up into sums[n][c][2];
 * pti_gn_bwd_apply then finishes dx = rstd*(gamma*dy - c1 - xhat*c2) [+ dres] and dgamma / dbeta +=.           */
int pti_conv_gnbwd_tiles(const pti_conv_desc* d);
/* Chained form (SURVEY 2.1 K4 "GroupNorm backward folded into the consumer's loader"): the launch's INPUT is not a
 * materialised gradient but the pair (g_in = dA * act'(GN(x_in)), x_in) of the GroupNorm ABOVE with its statistics,
 * weight and finalized sums {sum g, sum g*xhat} per (n, channel); the loader stages
 * rstd * (gamma * g - c1 - xhat * c2) -- what pti_gn_bwd_apply would have written -- and also writes it to dx_in_out
 * (bf16, same shape) for the weight gradient of the conv in between.  Everything else as pti_conv2d_mfma_gnbwd.
 * pti_conv_gnbwd_chain_supported(cin, cout, ksize): 3x3 with cin and cout multiples of 128.  The affine gradients of the
 * chained GroupNorm come from pti_gn_affine_grads (dgamma[c] += sum_n sums[n][c][1], dbeta[c] += sum_n sums[n][c][0]). */
int pti_conv_gnbwd_chain_supported(int cin, int cout, int ksize);
int pti_conv2d_mfma_gnbwd_chain(const void* g_in, const void* x_in, int x_in_f16, const int64_t* in_stats,
                                const float* in_gamma, const float* in_sums, void* dx_in_out, const void* w_packed,
                                const void* gx, const int64_t* gstats, const float* ggamma, const float* gbeta,
                                void* dy_out, float* gsums, const pti_conv_desc* d, int silu, pti_stream_t s);
int pti_gn_affine_grads(const float* sums, float* dgamma, float* dbeta, int n, int c, pti_stream_t s);
/* pti_gn_sums_finalize with pti_gn_affine_grads(affine_sums, dgamma, dbeta, n, affine_c) riding in the same launch */
int pti_gn_sums_finalize_affine(const float* partials, float* sums, int n, int tiles, int row_len, const float* affine_sums,
                                float* dgamma, float* dbeta, int affine_c, pti_stream_t s);
int pti_conv2d_mfma_gnbwd(const void* dy_in, const void* w_packed, const void* gx, const int64_t* gstats,
                          const float* ggamma, const float* gbeta, void* dy_out, float* gpartials,
                          const pti_conv_desc* d, int silu, pti_stream_t s);
int pti_gn_sums_finalize(const float* partials, float* sums, int n, int tiles, int row_len,
                         pti_stream_t s);
int pti_gn_bwd_apply(const void* x, const void* dy, const void* dres, void* dx, const int64_t* stats,
                     const float* gamma, const float* beta, const float* sums, float* dgamma,
                     float* dbeta, int n, int hw, int c, int groups, float eps, int x_f16,
                     pti_stream_t s);
/* y[n,h,w,c] = sum of the 2x2 block of x[n,2h,2w,c]: backward of nn.Upsample(nearest, 2x).     */
int pti_pool2x2_sum(const void* x, void* y, int n, int h, int w, int c, pti_stream_t s);

/* ---- mid-block self-attention (MONAI SpatialAttentionBlock -> SABlock, 1 head, dim = c) ------ */
/* qkv: bf16 [b,l,3c] (q|k|v per token, from the fused to_q/to_k/to_v 1x1 conv); o: bf16 [b,l,c] =
 * softmax(q k^T c^-0.5) v; lse2: fp32 [b,l] log2-sum-exp kept for the backward.  c in {64,128,256},
 * l % 64 == 0.  The l x l matrix is never materialised.                                          */
int pti_attention_fwd(const void* qkv, void* o, float* lse2, int b, int l, int c, pti_stream_t s);
/* dqkv: bf16 [b,l,3c] gradient w.r.t. qkv given dout (bf16 [b,l,c]); delta: fp32 [b,l] scratch.   */
int pti_attention_bwd(const void* qkv, const void* o, const void* dout, const float* lse2,
                      float* delta, void* dqkv, int b, int l, int c, pti_stream_t s);

/* ---- latent bottleneck (MONAI AutoencoderKL.encode tail / sampling / post_quant_conv) ------- */
/* h: fp32 [b,hw,l] (NHWC); eps/mu/sigma/logvar: fp32 [b,l,hw] (NCHW); zq: fp32 [b,hw,l].
 * eps NULL => deterministic z = mu.  Weights are the nn.Conv2d 1x1 masters [l][l], biases [l]. */
int pti_latent_head_fwd(const float* h, const float* eps, const float* wm, const float* bm,
                        const float* wl, const float* bl, const float* wp, const float* bp,
                        float* mu, float* sigma, float* logvar, float* zq, int b, int hw, int l,
                        pti_stream_t s);
int pti_post_quant(const float* z_nchw, const float* wp, const float* bp, float* zq_nhwc, int b,
                   int hw, int l, pti_stream_t s);
/* backward of pti_post_quant: dz (NCHW, may be NULL), gwp/gbp +=.  workspace: float scratch of
 * PTI_POST_QUANT_BWD_WS_FLOATS(l) elements (per-workgroup partials, added up in a fixed order by a second
 * launch: no float atomics, see pti_latent_head_bwd).                                             */
#define PTI_POST_QUANT_BWD_MAX_BLOCKS 256
#define PTI_POST_QUANT_BWD_WS_FLOATS(l) (PTI_POST_QUANT_BWD_MAX_BLOCKS * ((l) * (l) + (l)))
int pti_post_quant_bwd(const float* dzq_nhwc, const float* z_nchw, const float* wp, float* dz_nchw,
                       float* gwp, float* gbp, float* workspace, int b, int hw, int l,
                       pti_stream_t s);
/* backward of pti_latent_head_fwd: dh, and g* += the six 1x1-conv parameter gradients.  workspace: float scratch
 * of PTI_LATENT_BWD_WS_FLOATS(l) elements -- every workgroup stores its partial gradients there and a second
 * launch adds them up in a fixed order (no float atomics, bitwise reproducible: l == 4 -- the reference's latent
 * width -- keeps its partials in registers, other widths sum 128-element chunks in element order through LDS).   */
#define PTI_LATENT_BWD_MAX_BLOCKS 512
#define PTI_LATENT_BWD_WS_FLOATS(l) (PTI_LATENT_BWD_MAX_BLOCKS * 3 * ((l) * (l) + (l)))
int pti_latent_head_bwd(const float* h, const float* eps, const float* wm, const float* bm,
                        const float* wl, const float* bl, const float* wp, const float* bp,
                        const float* dzq, const float* dmu, const float* dsigma, float* dh,
                        float* gwm, float* gbm, float* gwl, float* gbl, float* gwp, float* gbp,
                        float* workspace, int b, int hw, int l, pti_stream_t s);

/* ---- loss step (reference src/pti_ldm_vae/models/losses.py:25-30,62-66; train_vae.py:393-394) */
/* out2[0] += mean recon loss (L1, or L2 when l2), out2[1] += mean_b KL; d_* receive the gradient
 * of recon + kl_weight*kl (NULL to skip).  third_mode 0: third used as log-variance (the
 * reference call site); 1: third is sigma, input_is_logvar=False semantics.  workspace: float scratch of
 * PTI_VAE_LOSS_WS_FLOATS elements (per-workgroup partials, added up in a fixed order by a second launch).   */
#define PTI_VAE_LOSS_MAX_BLOCKS 1024
#define PTI_VAE_LOSS_WS_FLOATS (2 * PTI_VAE_LOSS_MAX_BLOCKS)
int pti_vae_loss(const float* recon, const float* images, int64_t npix, const float* mu,
                 const float* third, int64_t nlat, int batch, float* out2, float* d_recon,
                 float* d_mu, float* d_third, float* workspace, int l2, int third_mode,
                 float kl_weight, pti_stream_t s);

/* ---- AR-VAE attribute regularisation (reference src/pti_ldm_vae/models/losses.py:69-166; train_vae.py:403-417) */
/* mu: fp32 [b,l,hw] (NCHW z_mu); attrs: fp32 [na,b] attribute values of the local batch; channels[q] / deltas[q]:
 * latent channel and tanh slope of attribute q; pair_mask: NULL = every ordered pair i != j ("all"), else uint8
 * [na,b,b] with mask[q][i][j] != 0 for the sampled pairs ("subset").  Writes per_attr[q] = mean over the selected pairs
 * with a_i != a_j of (tanh(delta (z_j - z_i)) - sign(a_j - a_i))^2 with z = mu.mean(hw) (0 when no pair qualifies) and
 * counts[q] = that number of pairs; d_mu (NULL to skip) += gamma * d(sum_q per_attr[q]) / d mu.  No atomics.        */
int pti_ar_vae_loss(const float* mu_nchw, int b, int l, int hw, const float* attrs, const int32_t* channels,
                    const float* deltas, int na, const uint8_t* pair_mask, float gamma, float* per_attr,
                    int32_t* counts, float* d_mu, pti_stream_t s);

/* ---- optimiser (torch.optim.Adam defaults, train_vae.py:301) on a flat fp32 arena ----------- */
int pti_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                  float beta2, float eps, int step, float grad_scale, pti_stream_t s);

/* ---- layout casts at the model boundary ------------------------------------------------------ */
int pti_cast_nchw_f32_to_nhwc_bf16(const float* x, void* y, int n, int c, int hw, pti_stream_t s);
int pti_cast_nhwc_bf16_to_nchw_f32(const void* x, float* y, int n, int c, int hw, pti_stream_t s);

/* ---- input pipeline ---------------------------------------------------- */
/* Batch form of the reference's per-sample transform chain Resize(patch_size) [MONAI default mode "area" =     *
 * adaptive average pooling] -> LocalNormalizeByMask -> float32 (data/dataloaders.py:319-329,                    *
 * data/transforms.py:8-32).  src: the raw fp32 images of the batch concatenated; offsets[b], hw[b] = {H, W}:    *
 * where image b starts and its size (device arrays); out: [b][1][hp][wp] fp32; stats: device scratch of 3*b     *
 * doubles (zeroed here).                                                                                        */
int pti_preprocess_batch(const float* src, const int64_t* offsets, const int32_t* hw, int b, int hp, int wp,
                         float* out, double* stats, pti_stream_t s);

/* ---- PatchDiscriminator + adversarial loss --------------------------------------------------------
 * Reference: vae_scripts/train_vae.py:266-279 (MONAI PatchDiscriminator(spatial_dims=2, num_layers_d=3, channels=32,
 * in_channels=1, out_channels=1, norm="INSTANCE")), :298 (PatchAdversarialLoss("least_squares")), :399-401 (generator
 * term), :447-458 (discriminator step).  A 4x4 convolution is lowered to patches x 1x1 convolution: the product, its
 * data gradient and its weight gradient run on pti_conv2d_mfma / pti_conv_wgrad_mfma (ksize 1); the entry points below
 * are the discriminator-specific passes.  Patches: bf16 [n][ho][wo][16*c], column = (ky*4 + kx)*c + channel,
 * ho = (h + 2 - 4)/stride + 1.  Normalisation tables: float [n][c][2] = {mean, rstd} (InstanceNorm2d, biased variance).
 * c in {32, 64, 128, 256}.  No floating-point atomics: results are bitwise reproducible.                              */
/* fp32 image [n][h][w] (1 channel) -> patches [n][h/2][w/2][32]: 16 taps of the 4x4 stride-2 pad-1 window + 16 zeros. */
int pti_pd_im2col_image(const float* img, void* patches, int n, int h, int w, pti_stream_t s);
/* bf16 [n][h][w][c] -> patches of act(norm(src)): norm optional ({mean,rstd} table), act = LeakyReLU(slope) if `act`. */
int pti_pd_im2col(const void* src, const float* norm, void* patches, int n, int h, int w, int c, int stride,
                  int act, float slope, pti_stream_t s);
/* nn.InstanceNorm2d statistics of y bf16 [n][hw][c] -> table [n][c][2] = {mean, 1/sqrt(var + eps)}. */
int pti_pd_in_stats(const void* y, float* table, int n, int hw, int c, float eps, pti_stream_t s);
/* Data gradient of pti_pd_im2col fused with LeakyReLU'(norm(y_prev)): g = act'(xhat) * col2im(d_patches), bf16
 * [n][h][w][c]; with a norm table also writes the InstanceNorm-backward block partials {sum g, sum g*xhat}:
 * partials float [n][pti_pd_col2im_blocks(n, h*w, c)][c][2], to be summed by pti_gn_sums_finalize(row_len = 2c).      */
int pti_pd_col2im_blocks(int n, int hw, int c);
int pti_pd_col2im(const void* d_patches, const void* y_prev, const float* norm, void* g, float* partials, int n,
                  int h, int w, int c, int stride, float slope, pti_stream_t s);
/* Gradient w.r.t. the 1-channel image from d_patches [n][h/2][w/2][32]: d_img = (accumulate ? d_img : 0) + scale * sum. */
int pti_pd_col2im_image(const void* d_patches, float* d_img, int n, int h, int w, float scale, int accumulate,
                        pti_stream_t s);
/* InstanceNorm backward, second pass: dy = rstd * (g - sums[0]/hw - xhat * sums[1]/hw); sums float [n][c][2]; dy may be g. */
int pti_pd_in_bwd_apply(const void* g, const void* y, const float* norm, const float* sums, void* dy, int n, int hw,
                        int c, pti_stream_t s);
/* PatchAdversarialLoss(criterion="least_squares"): loss_out[0] = mean((LeakyReLU_slope(logit) - target)^2) over `count`
 * 16-bit logits (element m at logits[m*stride]; slope 0.05 = MONAI's default activation, 1 = none).  loss_out must hold
 * 1 + pti_pd_lsgan_blocks(count) floats (block partials, summed in block order by a second launch).  When d_logits is
 * given, row m of bf16 [count][stride] = {grad_scale * (a_m - target) * LeakyReLU'(logit_m), 0, ...}: pass
 * grad_scale = weight * 2 / count for d(weight * loss)/d logits.                                                     */
int pti_pd_lsgan_blocks(int count);
int pti_pd_lsgan(const void* logits, int logits_f16, int stride, int count, float target, float slope,
                 float grad_scale, float* loss_out, void* d_logits, pti_stream_t s);

/* Final block of the discriminator (C -> 1 channel, 4x4, stride 1, pad 1) WITHOUT a patch matrix: with one output
 * channel the lowering above would move ~15x the bytes of its input.  The activated input LeakyReLU(norm(y_prev)) is
 * rebuilt on the fly; logits and d_logits are fp32 [n][h-1][w-1]; w is fp32 [16][c] (tap-major), bias fp32 [1].
 *   fwd:   logits = conv(act(norm(y_prev)), w) + bias
 *   dgrad: g = act'(xhat) * conv^T(d_logits, w)  (bf16 [n][h][w][c]) + InstanceNorm-backward partials as pti_pd_col2im
 *          (same pti_pd_col2im_blocks(n, h*w, c) rows)
 *   wgrad: partials float [pti_pd_final_wgrad_blocks(n,h,w)][16*c + 8] = per-block {dw[16][c], dbias, 0..}; sum the rows
 *          in block order (pti_gn_sums_finalize(partials, out, 1, blocks, 16*c + 8)).
 * pti_pd_lsgan accepts such fp32 logits with logits_f16 = 2 (stride in floats, d_logits then is float[count*stride]). */
int pti_pd_final_fwd(const void* y_prev, const float* norm, const float* w, const float* bias, float* logits, int n,
                     int h, int w_, int c, float slope, pti_stream_t s);
int pti_pd_final_dgrad(const float* d_logits, const void* y_prev, const float* norm, const float* w, void* g,
                       float* partials, int n, int h, int w_, int c, float slope, pti_stream_t s);
int pti_pd_final_wgrad_blocks(int n, int h, int w_);
int pti_pd_final_wgrad(const float* d_logits, const void* y_prev, const float* norm, float* partials, int n, int h,
                       int w_, int c, float slope, pti_stream_t s);

/* ---- LPIPS comparison tail of the perceptual term ----------------------------------------------------
 * Reference: vae_scripts/train_vae.py:299, :395-397 (monai.losses.PerceptualLoss(spatial_dims=2, network_type="squeeze")
 * = lpips.LPIPS(net="squeeze"): per feature tap normalize_tensor on both maps, squared difference, the 1x1 `lin` layer,
 * spatial mean).  a, b: fp32 NCHW feature maps [n][c][hw] of the reconstruction / the target, w: fp32 [c] (the lin
 * layer's weight).  The feature network itself stays torch ops (models/perceptual.py); this is its memory-bound tail.
 *   fwd: partials float [n][pti_lpips_tap_blocks(c, hw)] -- per-workgroup sums of sum_c w_c (a_c/(|a|+1e-10) -
 *        b_c/(|b|+1e-10))^2 over their pixels; the tap's value for sample i is sum(partials[i][:]) / hw.
 *        saved float [n][3][hw] = {|a_p|, |b_p|, sum_c w_c d_c a_c}: what the backward needs per pixel.
 *   bwd: ga [n][c][hw] = gout[i] * d value_i / d a   (gout fp32 [n]); b and w get no gradient (frozen network, target).
 * No atomics: bitwise reproducible.                                                                                   */
int pti_lpips_tap_blocks(int c, int hw);
int pti_lpips_tap_fwd(const float* a, const float* b, const float* w, float* saved, float* partials, int n, int c, int hw,
                      pti_stream_t s);
int pti_lpips_tap_bwd(const float* a, const float* b, const float* w, const float* saved, const float* gout, float* ga,
                      int n, int c, int hw, pti_stream_t s);

/* ---- trunk of the perceptual network: the passes between its convolutions ---------------------------
 * Reference: torchvision squeezenet1_1.features under lpips.LPIPS(net="squeeze") (train_vae.py:299): Fire modules and
 * MaxPool2d(kernel 3, stride 2, ceil_mode=True).  The Fire convolutions run on pti_conv2d_mfma (fp16 forward, bf16
 * data gradient); activations NHWC fp16, gradients NHWC bf16, element counts / channels multiples of 8.
 *   pti_relu_f16:        x = max(x, 0) in place.
 *   pti_relu_bwd:        g = y > 0 ? g : 0 in place (y = the ReLU OUTPUT).
 *   pti_relu_bwd_add:    g = y > 0 ? g + g2 : 0 in place (a tap's own gradient + the one arriving from later layers).
 *   pti_maxpool3s2_out:  pooled size of one spatial dimension.
 *   pti_maxpool3s2_fwd:  y [n][ho][wo][c] = max over the (clipped) 3x3 windows of x [n][h][w][c].
 *   pti_maxpool3s2_bwd:  gx (+)= gather of gy over the windows whose maximum the element is (no atomics).
 *   pti_nchw_f32_to_nhwc_f16 / pti_nhwc_bf16_add_to_nchw_f32: the trunk's boundary with torch's layout (tap 0):
 *                        y[n][p][c] = (fp16) x[n][c][p];   y[n][c][p] += (float) g[n][p][c];   c a multiple of 64.
 *   pti_lpips_tap_nhwc_*: pti_lpips_tap_* on the trunk's layout -- a, b fp16 [n][hw][c], ga bf16 [n][hw][c]; same
 *                        `saved` / `partials` contract with pti_lpips_tap_nhwc_blocks(c, hw); c = 8 * L * k with
 *                        L in {8, 16, 32, 64}, k <= 4 (0 blocks = unsupported channel count).
 *   pti_squeeze_conv1_*: the first layer (3 -> 64, 3x3, stride 2, no padding, + ReLU) for a ONE-channel image whose three
 *                        copies the reference scales per channel (ensure_three_channels + lpips ScalingLayer), folded
 *                        into a 1 -> 64 convolution: w10 fp32 [10][64] = {W'[tap][co] = sum_c W[co][c][tap] / scale_c,
 *                        b'[co] = b[co] - sum_c shift_c / scale_c * sum_tap W[co][c][tap]}.  x fp32 [n][h][w];
 *                        fwd: y fp16 [n][(h-3)/2+1][(w-3)/2+1][64] = tap 0;  bwd: dx fp32 [n][h][w] from the bf16
 *                        gradient g w.r.t. tap 0 (ReLU mask taken from t0 = y; t0 = NULL: g is already masked).           */
int pti_relu_f16(void* x, int64_t count, pti_stream_t s);
int pti_relu_bwd(void* g, const void* y, int64_t count, pti_stream_t s);
int pti_relu_bwd_add(void* g, const void* g2, const void* y, int64_t count, pti_stream_t s);
int pti_maxpool3s2_out(int h);
int pti_maxpool3s2_fwd(const void* x, void* y, int n, int h, int w, int c, pti_stream_t s);
int pti_maxpool3s2_bwd(const void* gy, const void* x, const void* y, void* gx, int n, int h, int w, int c, int accumulate,
                       pti_stream_t s);
int pti_squeeze_conv1_fwd(const float* x, const float* w10, void* y, int n, int h, int w, pti_stream_t s);
int pti_squeeze_conv1_bwd(const void* g, const void* t0, const float* w10, float* dx, int n, int h, int w, pti_stream_t s);
int pti_nchw_f32_to_nhwc_f16(const float* x, void* y, int n, int c, int hw, pti_stream_t s);
int pti_nhwc_bf16_add_to_nchw_f32(const void* g, float* y, int n, int c, int hw, pti_stream_t s);
/* Image-side boundary of the MFMA path for 2..8-channel images (csrc/narrow_pad.hip; replaces nothing in the reference:
 * conv_in / conv_out of MONAI's AutoencoderKL, src/pti_ldm_vae/models/autoencoder.py:67-79, take [N,C,H,W] fp32 images):
 *   pti_pad_nchw_to_nhwc32:   ya, yb [n][hw][32] 16-bit (fp16 if *_f16 else bf16; yb may be NULL) <- x [n][c][hw] fp32,
 *                             channels >= c written as zeros; 1 <= c <= 8.
 *   pti_slice_nhwc32_to_nchw: y [n][c][hw] fp32 <- the first c channels of x [n][hw][32] (fp16 if x_f16 else bf16). */
int pti_pad_nchw_to_nhwc32(const float* x, void* ya, void* yb, int n, int c, int hw, int a_f16, int b_f16, pti_stream_t s);
int pti_slice_nhwc32_to_nchw(const void* x, float* y, int n, int c, int hw, int x_f16, pti_stream_t s);
int pti_lpips_tap_nhwc_blocks(int c, int hw);
int pti_lpips_tap_nhwc_fwd(const void* a, const void* b, const float* w, float* saved, float* partials, int n, int c,
                           int hw, pti_stream_t s);
int pti_lpips_tap_nhwc_bwd(const void* a, const void* b, const float* w, const float* saved, const float* gout, void* ga,
                           int n, int c, int hw, pti_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* PTI_VAE_H */
