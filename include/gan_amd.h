/* gan_amd.h — C ABI of the MI355X (gfx950) Pix2Pix / CycleGAN training hot path.
 *
 * The reference (kingjosephm/GAN) has no FFI / plugin interface: its hot path is Python calling
 * tf.keras layers.  Each entry point below replaces one Keras layer invocation
 * (or its autodiff counterpart) that the reference instantiates; the reference file:line it
 * replaces is cited per function.  INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *  - plain C, raw device pointers, POD structs; no torch / hip types in signatures
 *    (gan_stream_t is a hipStream_t passed as void*).
 *  - every call only ENQUEUES work on `stream` (hipGraph-capturable, no allocation, no sync).
 *  - return: 0 ok; <0 invalid argument / unsupported shape (GAN_E_*); >0 a hipError_t.
 *  - activations NHWC; a GanTensor may be a channel slice of a wider buffer (pitch > c), which is
 *    how skip-concat (base_gan.py:206,221) and the discriminator input concat (base_gan.py:139)
 *    are realised without a copy.
 *  - dtype GAN_F32: fp32 storage, exact-fp32 MFMA (v_mfma_f32_16x16x4_f32) — the parity path.
 *    dtype GAN_BF16: bf16 storage, v_mfma_f32_16x16x32_bf16, fp32 accumulate — the fast path.
 *    dtype GAN_F16: fp16 storage, v_mfma_f32_16x16x32_f16, fp32 accumulate and fp32 master weights; used with the
 *    dynamic loss scale below (BASELINE.json config 5; the reference itself has no mixed precision).
 *  - every descriptor struct starts with `uint32_t struct_size` = sizeof(the struct) as the CALLER compiled it; an entry
 *    point returns GAN_E_ARG when it differs from the library's own sizeof (descriptors grow by appending fields).
 *  - convolution weights are consumed in "NK" layout [16 taps][rows][k] (k contiguous), produced
 *    from the fp32 Keras-layout master by gan_weights_prepare().
 */
#ifndef GAN_AMD_H
#define GAN_AMD_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* gan_stream_t;

enum { GAN_F32 = 0, GAN_BF16 = 1, GAN_F16 = 2 };
enum { GAN_ACT_NONE = 0, GAN_ACT_LRELU = 1, GAN_ACT_RELU = 2, GAN_ACT_TANH = 3 };
enum { GAN_E_ARG = -1, GAN_E_SHAPE = -2, GAN_E_WORKSPACE = -3 };

typedef struct GanTensor {
  void* ptr;          /* element (n=0,h=0,w=0,channel 0 of this view) */
  int32_t n, h, w, c; /* c: channels the op touches */
  int32_t pitch;      /* elements between consecutive pixels (>= c) */
} GanTensor;

/* ---- convolutions ------------------------------------------------------------------------- */
/* Optional fused backward epilogue of the two dgrad entry points: the backward of the layer BELOW (its
 * normalisation + activation, base_gan.py:83-87 / :113-120 under tf.GradientTape) starts in the epilogue of the
 * launch that produces the gradient w.r.t. that layer's activation.  For the first `cols` channels of y the stored
 * value is not da but
 *     dz = (da + add) * act'(z) * 2*dropmask,   z = gamma*(ref - mean)*rstd + beta      (mean != NULL; ref = that layer's y)
 *     dz = (da + add) * act'(ref)                                                       (mean == NULL; ref = saved activation)
 * and, with mean != NULL, per-tile partial sums (sum dz, sum dz*xhat) are written to GanConvDesc.stats_partial as
 * [group][chunk][cols][2] (groups = GanConvDesc.stats_groups); gan_norm_act_bwd_fused() finishes the layer.  Channels
 * >= cols (the skip half of a decoder concat) are stored unchanged.  Honoured only when gan_conv_plan_info()[4] > 0. */
typedef struct GanBwdFuse {
  GanTensor ref;             /* y (pre-normalisation) or the saved activation a; same n, h, w as GanConvDesc.y */
  GanTensor add;             /* optional second upstream gradient (skip connection); ptr NULL = none */
  const float* mean;         /* [groups][cols]; NULL = activation-only backward */
  const float* rstd;
  const float* gamma;        /* [cols] */
  const float* beta;
  const uint8_t* dropmask;   /* optional dense 0/1 bytes [n][h][w][mask_pitch] */
  int32_t mask_pitch;
  int32_t act;               /* GAN_ACT_LRELU | GAN_ACT_RELU */
  float slope;
  int32_t cols;              /* multiple of 8 */
} GanBwdFuse;

/* Optional: a split-K launch of a SMALL layer finishes the whole layer in its slab-reduce kernel (one launch instead of
 * reduce + statistics finalize + apply): a workgroup owns 8 channels of all rows of a statistics group (at most 1024 rows), so
 * it sums the K slabs, takes the statistics and applies them without leaving the kernel.
 * Forward entry points (with GanConvDesc.stats_groups): y = conv output as usual, out = act(dropout(gamma*(y-mean)*rstd+beta))
 * (Conv -> BN|IN -> [Dropout] -> act of base_gan.py:77-87, :106-120); mean / rstd / moving averages are written as
 * gan_norm_stats would.  dgrad entry points (with GanConvDesc.bwd_fuse, which names the layer below): out = dy of that layer,
 * dgamma / dbeta accumulated; the first bwd_fuse->cols channels of GanConvDesc.y are then NOT written (dz stays in registers;
 * channels >= cols receive the plain gradient as usual).
 * Honoured only when gan_conv_plan_info()[4] == -1: the caller asks first and passes norm_fuse = NULL otherwise - a launch
 * whose plan cannot honour a non-NULL norm_fuse returns GAN_E_SHAPE (the caller would have dropped the layer's own
 * normalisation launches on the strength of the query). */
typedef struct GanNormFuse {
  GanTensor out;
  const float* gamma;
  const float* beta;
  float* mean;               /* forward: [groups][c] written */
  float* rstd;
  float* moving_mean;        /* forward, optional */
  float* moving_var;
  float eps, momentum;
  const uint8_t* dropmask;   /* forward, optional: dense 0/1 bytes [n][h][w][c] */
  int32_t act;
  float slope;
  float* dgamma;             /* backward, optional */
  float* dbeta;
  int32_t accumulate;
} GanNormFuse;

typedef struct GanConvDesc {
  uint32_t struct_size;  /* sizeof(GanConvDesc) of the caller's build */
  int32_t dtype;
  int32_t stride;        /* Conv2D: 1 or 2 (k4, zero pad 1 each side); Conv2DTranspose: 2 */
  GanTensor x;           /* GEMM-K side tensor; x.c must be a multiple of 8 (zero-padded channels) */
  GanTensor y;           /* produced tensor; y.c = channels written (pitch may be larger) */
  const void* w;         /* NK weights [16][w_rows][x.c], dtype */
  int32_t w_rows;        /* >= y.c */
  const float* bias;     /* optional [y.c] */
  int32_t act;           /* epilogue activation (GAN_ACT_*) applied after bias */
  float slope;           /* LeakyReLU alpha */
  int32_t y_f32;         /* 1: write y as fp32 whatever dtype is (used for logits) */
  void* workspace;       /* split-K slabs; size from gan_conv_workspace_bytes() */
  size_t workspace_bytes;
  float* stats_partial;  /* optional: fused normalisation statistics — per-tile (sum, sum^2) partials of y,   */
  int32_t stats_groups;  /* laid out [group][chunk][y.c][2]; emitted only if gan_conv_plan_info()[4] > 0     */
  size_t stats_partial_bytes; /* size of that region: groups * chunks * channels * 8 bytes are written (GAN_E_WORKSPACE if short) */
  const GanBwdFuse* bwd_fuse; /* optional (dgrad entry points only), see above */
  const GanNormFuse* norm_fuse; /* optional, see above */
} GanConvDesc;

/* Conv2D(k4, strides=s, no bias | bias) — base_gan.py:77-79 ('same', s=2), :145-148 and :157-161
 * (ZeroPadding2D + 'valid', s=1).  w = prepared from HWIO master with gan_weights_prepare(transposed). */
int gan_conv2d_fwd(const GanConvDesc* d, gan_stream_t stream);
/* Gradient of the above w.r.t. its input (tf.GradientTape, pix2pix.py:210-211): x := dy, y := dx,
 * w = native NK copy of the HWIO master ([tap][cin][cout]). */
int gan_conv2d_dgrad(const GanConvDesc* d, gan_stream_t stream);
/* Conv2DTranspose(k4, strides=2, 'same') — base_gan.py:106-110, :201-204 (bias + tanh head).
 * w = native NK copy of the (kh,kw,cout,cin) master. */
int gan_convT2d_fwd(const GanConvDesc* d, gan_stream_t stream);
/* Gradient of Conv2DTranspose w.r.t. its input: x := dy (2h x 2w), y := dx (h x w),
 * w = transposed NK copy ([tap][cin][cout]). */
int gan_convT2d_dgrad(const GanConvDesc* d, gan_stream_t stream);
size_t gan_conv_workspace_bytes(const GanConvDesc* d, int op /*0 conv_fwd,1 conv_dgrad,2 convT_fwd,3 convT_dgrad*/);
/* launch plan the library picks for this problem: info[5] = {BM, BN, splitK, parities, fused-stats chunks per group}
 * (BM = 0: streaming kernel family BN = 1 | 2, or BN = 8: the column-owner kernel - one launch, 8 output channels of every row per
 * workgroup, only with norm_fuse; BM = 1024: parity-patch kernel; info[4] = -1: norm_fuse is honoured, the launch finishes the layer) */
int gan_conv_plan_info(const GanConvDesc* d, int op, int32_t* info);
/* Which main loop a 256-row tile of this launch takes: 0 = one staged A tile per tap (conv_gemm_pp_kernel), 2 | 4 = the taps of a
 * kernel row that read the same pixels one grid column apart share one staged A tile (conv_gemm_ps_kernel; option conv.tap_share).
 * 0 for every other tile; < 0: error code.  Results of the two loops differ in fp32 summation order only. */
int gan_conv_tap_shared(const GanConvDesc* d, int op);

/* ---- layer stack: a run of consecutive small split-K layers in ONE launch -------------------------------------------------
 * The inner layers of the U-Net (base_gan.py:183-193: down5..down8, up1..up3 at batch 16; most of the generator at the reference's
 * CycleGAN batch of 1) are latency chains: a GEMM into fp32 slabs plus the slab reduce that finishes the layer (GanNormFuse), each a
 * few microseconds of work behind a launch.  gan_conv_stack_* runs such a run of layers from ONE persistent kernel: a resident grid
 * walks the layers, a grid barrier separates the phases, and what crosses workgroups inside the launch is written through / read
 * around the per-XCD L2s.  Eligible: a launch whose gan_conv_plan_info()[4] == -1 (norm_fuse honoured), tile 64x128 or 16x128, at
 * most 512 rows per statistics group; layer i+1 must consume what layer i produced (the caller's op order), all one dtype.
 *   plan   : host side, once per run of layers -> an opaque blob (gan_conv_stack_plan_bytes(n) bytes) that the caller ALSO copies
 *            into device memory (256-byte aligned) - the library allocates nothing;
 *   launch : enqueue-only (hipGraph-capturable).  barrier_state: gan_conv_stack_barrier_bytes() of ZEROED device memory, 128-byte
 *            aligned, owned by this plan for good (monotonic counters); err_flag: device int32, set to 1 if a grid barrier timed out
 *            (~2 s: the grid could not become resident) - results are then undefined, the kernel still terminates.
 * Measured (round 4, tools/bench_stack.py, hipGraph replay on one stream): 5 layers at Pix2Pix batch 16 89.6 us as a stack against
 * 79.3 us as ten launches; 9 layers at CycleGAN batch 1 141 against 122 us - two grid barriers (2.2 us each + the drain of the
 * write-through stores) cost more than the two kernel boundaries of a replayed graph (~1.5 us each), and every layer keeps its four
 * dependent global round trips.  The in-tree callers therefore use it only on request (option conv.stack).
 * At most TWO stack launches may run concurrently on one GPU (two workgroups of it fit on every CU; a third resident grid could
 * wait for CUs the other two hold while they wait for each other). */
int gan_conv_stack_eligible(const GanConvDesc* d, int op);   /* 1: this launch can be a layer of a stack, 0: not, < 0: error code */
size_t gan_conv_stack_plan_bytes(int32_t n);
int gan_conv_stack_plan(const GanConvDesc* const* descs, const int32_t* ops /* as gan_conv_plan_info's op */, int32_t n, void* host_plan,
                        size_t plan_bytes);
int gan_conv_stack_launch(const void* host_plan, const void* dev_plan, void* barrier_state, int32_t* err_flag, gan_stream_t stream);
size_t gan_conv_stack_barrier_bytes(void);

/* Optional: the wgrad launch itself applies the optimiser step (TF-form Adam, base_gan.py:247-252 / pix2pix.py:213-216) to the
 * kernel it has just differentiated and refreshes its typed NK copies - the fp32 gradient is then neither written nor read back
 * (dw stays untouched).  Bit-identical to gan_conv_wgrad followed by gan_adam_prepare_multi.  gan_adam_begin must have run for
 * this step (lr_t).  Not for fp16 steps with dynamic loss scaling (their update waits for the whole-step inf/nan check and
 * un-scales).  Two carriers: the epilogue of an un-split launch of the 128x128 LDS-DMA kernel, and - for every launch that
 * splits its reduction, the ping-pong kernel included - the slab-reduce kernel, which then ends in the optimiser step instead of
 * writing the fp32 gradient (option wgrad.reduce_adam, default 1; kernels of at least
 * wgrad.reduce_adam_min_params parameters).  Honoured only when gan_wgrad_adam_fused() returns 1 for the
 * descriptor (16-bit storage, accumulate == 0, channel counts multiples of 8, no tap folding); the caller asks first and passes adam_fuse = NULL
 * otherwise - gan_conv_wgrad returns GAN_E_SHAPE for a non-NULL adam_fuse its plan cannot honour. */
typedef struct GanAdamFuse {
  float* master;             /* fp32 [16][big_c][small_c]: the layout of dw */
  float* m;
  float* v;
  void* nk_native;           /* [16][big_c][small_c] dtype, or NULL */
  void* nk_transposed;       /* [16][small_c][big_c] dtype, or NULL */
  const float* lr_t;         /* device scalar written by gan_adam_begin */
  float beta1, beta2, eps;
} GanAdamFuse;

typedef struct GanWgradDesc {
  uint32_t struct_size;  /* sizeof(GanWgradDesc) of the caller's build */
  int32_t dtype;
  int32_t stride;        /* 1 or 2 */
  GanTensor big;         /* tensor on the fine grid  (Conv2D: layer input x;  Conv2DTranspose: dy) */
  GanTensor small;       /* tensor on the coarse grid (Conv2D: dy;            Conv2DTranspose: layer input x) */
  float* dw;             /* fp32 [16][big_c][small_c]: HWIO for Conv2D, (kh,kw,cout,cin) for Conv2DTranspose; 16-byte aligned */
  int32_t big_c, small_c;/* real channel counts written (<= big.c, small.c which are 8-padded) */
  int32_t accumulate;    /* 1: dw += result (a net called several times per step, cycle_gan.py:252-255) */
  void* workspace;
  size_t workspace_bytes;
  int32_t concurrent;    /* scheduling hint: nonzero = this launch shares the GPU with other streams' kernels (a side lane of a
                            captured step): the planner then prefers fewer, longer blocks (less slab traffic; the other streams get
                            the rest of the chip).  2 = beside a MIRROR chain doing the same work (the two-chain CycleGAN step):
                            the K-split target is halved as well. */
  const GanAdamFuse* adam_fuse; /* optional, see above */
  void* dw_wire;         /* optional (data-parallel steps, no counterpart in the single-device reference): the bfloat16 wire buffer of the
                            gradient exchange at this kernel's offset ([16][big_c][small_c], 8-byte aligned).  When gan_wgrad_wire_direct()
                            returns 1 the launch writes the gradient there in the wire format (gan_grad_pack's rounding) and leaves dw
                            untouched: no fp32 gradient, no cast pass.  Not together with adam_fuse / accumulate; a launch whose plan
                            cannot honour a non-NULL dw_wire returns GAN_E_SHAPE. */
} GanWgradDesc;
/* Kernel gradient of Conv2D / Conv2DTranspose (GradientTape.gradient w.r.t. trainable_variables,
 * pix2pix.py:210-211, cycle_gan.py:252-260). */
int gan_conv_wgrad(const GanWgradDesc* d, gan_stream_t stream);
size_t gan_wgrad_workspace_bytes(const GanWgradDesc* d);
int gan_wgrad_plan_info(const GanWgradDesc* d, int32_t* info /* {TA, TB, splitM, fold} */);
int gan_wgrad_adam_fused(const GanWgradDesc* d);   /* 1: d->adam_fuse will be honoured, 0: not, < 0: error code */
int gan_wgrad_wire_direct(const GanWgradDesc* d);  /* 1: d->dw_wire will be honoured, 0: not, < 0: error code */

/* Produce typed NK copies from a fp32 Keras-layout master [16][A][B]:
 * nk_native [16][A][pad8(B)] and nk_transposed [16][B][pad8(A)] (either may be NULL). */
int gan_weights_prepare(const float* master, int32_t A, int32_t B, int32_t dtype,
                        void* nk_native, void* nk_transposed, gan_stream_t stream);

/* The same for every kernel of a network in one launch.  entries_dev: device array of n GanPrepEntry; tile_start =
 * running sum of 16 * ceil(pad8(A)/64) * ceil(pad8(B)/64) over the preceding entries, tiles_b = ceil(pad8(B)/64). */
typedef struct GanPrepEntry {
  const float* master; void* nk_native; void* nk_transposed;
  int32_t A, B, tile_start, tiles_b;
} GanPrepEntry;
int gan_weights_prepare_multi(const void* entries_dev, int32_t n, int32_t total_tiles, int32_t dtype, gan_stream_t stream);
/* Keras Adam (base_gan.py:247-252) fused with the above for the kernel tensors of a network: every listed tensor (its
 * master pointer must lie inside the flat `master` buffer; m / v / grad are indexed at the same offset) is updated in
 * TF form and its NK copies rewritten in the same pass.  gan_adam_begin must have run this step; non-kernel parameters
 * (norm scales/offsets, biases) are updated with gan_adam_tf.  All four buffers 16-byte aligned. */
int gan_adam_prepare_multi(const void* entries_dev, int32_t n, int32_t total_tiles, int32_t dtype, float* master, float* m,
                           float* v, const void* grad, const float* lr_t, float beta1, float beta2, float eps,
                           float grad_scale, const float* scale_state, int32_t grad_bf16, gan_stream_t stream);

/* ---- normalisation + activation ------------------------------------------------------------- */
typedef struct GanNormDesc {
  uint32_t struct_size;  /* sizeof(GanNormDesc) of the caller's build */
  int32_t dtype;
  GanTensor y;              /* raw convolution output */
  GanTensor a;              /* out: act(dropout(gamma * (y - mean) * rstd + beta)) */
  int32_t groups;           /* statistics groups over the batch dimension: 1 = BatchNormalization over the
                               whole batch (base_gan.py:83,113,151); y.n = InstanceNormalization
                               (utils.py:26-30); 2 = two BatchNormalization calls (D(real), D(fake),
                               pix2pix.py:202-203) batched into one launch */
  float eps;                /* 1e-3 Keras BN default; 1e-5 utils.py:9 */
  const float* gamma;       /* [c] gamma / scale */
  const float* beta;        /* [c] beta / offset */
  float* mean;              /* [groups][c] */
  float* rstd;              /* [groups][c] */
  float* moving_mean;       /* optional [c]: BN moving averages, updated once per group in order */
  float* moving_var;
  float momentum;           /* 0.99 */
  const uint8_t* dropmask;  /* optional [n*h*w*c] 0/1; survivors x2 (Dropout(0.5), base_gan.py:117-118) */
  int32_t act;
  float slope;
  void* workspace;          /* >= gan_norm_workspace_bytes() */
  size_t workspace_bytes;
  uint32_t* sync;           /* optional: gan_norm_sync_bytes() of device memory, zero when first used and owned by this
                               layer invocation (launches that share an area must be stream-ordered; the area zeroes
                               itself again): lets gan_norm_finalize_act_fwd() run as ONE launch */
} GanNormDesc;
int gan_norm_stats(const GanNormDesc* d, gan_stream_t stream);
/* finalize only: d->workspace already holds `chunks` partials per group written by a convolution epilogue */
int gan_norm_stats_finalize(const GanNormDesc* d, int32_t chunks, gan_stream_t stream);
int gan_norm_act_fwd(const GanNormDesc* d, gan_stream_t stream);
size_t gan_norm_workspace_bytes(int32_t groups, int32_t c, int64_t rows_per_group);
/* The same Keras layer call (base_gan.py:83-87, 113-120, 151-155; utils.py:26-30) in two steps whose second one is a single
 * launch: gan_norm_stats_partial() = the partial sums of gan_norm_stats() without its finalize launch;
 * gan_norm_finalize_act_fwd() = gan_norm_stats_finalize(d, chunks) + gan_norm_act_fwd(d), bit-identical to that pair
 * (chunks <= 0: the partials gan_norm_stats_partial() wrote).  With d->sync the finalize rides inside the apply launch:
 * its first workgroups finalize and publish mean / rstd (+ the moving averages), the others wait for them with their rows'
 * loads already in flight (option norm.fin_in_apply, default 1; without a sync area, or when the apply grid has fewer
 * workgroups than the finalize has work units: two launches).  Every wait is bounded; gan_norm_sync_error_offset() locates the flag a
 * timed-out wait leaves in the area. */
int gan_norm_stats_partial(const GanNormDesc* d, gan_stream_t stream);
int gan_norm_finalize_act_fwd(const GanNormDesc* d, int32_t chunks, gan_stream_t stream);
size_t gan_norm_sync_bytes(void);
/* byte offset, inside a sync area, of the 32-bit word that is non-zero after a wait timed out (host reads it after a sync) */
size_t gan_norm_sync_error_offset(void);

typedef struct GanNormBwdDesc {
  uint32_t struct_size;  /* sizeof(GanNormBwdDesc) of the caller's build */
  int32_t dtype;
  GanTensor y;              /* raw convolution output saved by forward */
  GanTensor da;             /* upstream gradient w.r.t. a */
  GanTensor da2;            /* optional second upstream (ptr NULL if absent): skip-connection gradient */
  GanTensor dy;             /* out: gradient w.r.t. y */
  int32_t groups;
  const float* gamma;
  const float* beta;
  const float* mean;
  const float* rstd;
  const uint8_t* dropmask;
  int32_t act;
  float slope;
  float* dgamma;            /* optional [c] */
  float* dbeta;
  int32_t accumulate;       /* dgamma/dbeta += */
  void* workspace;
  size_t workspace_bytes;
  uint32_t* sync;           /* optional sync area (GanNormDesc.sync): the apply launch of gan_norm_act_bwd() /
                               gan_norm_act_bwd_fused() then also finalizes the sums, dgamma and dbeta (one launch less) */
} GanNormBwdDesc;
int gan_norm_act_bwd(const GanNormBwdDesc* d, gan_stream_t stream);
/* The same layer backward when its first half already ran in the producing dgrad's epilogue (GanBwdFuse above): d->da holds
 * dz (da2, dropmask, act are ignored / must be NULL), d->workspace the producer's partial sums, `chunks` per group as
 * reported by gan_conv_plan_info()[4], followed by room for [groups][c][2] floats.  Finalize + apply: dy, dgamma, dbeta. */
int gan_norm_act_bwd_fused(const GanNormBwdDesc* d, int32_t chunks, gan_stream_t stream);

typedef struct GanActBwdDesc { /* layers without normalisation: conv -> [bias] -> act */
  uint32_t struct_size;  /* sizeof(GanActBwdDesc) of the caller's build */
  int32_t dtype;
  GanTensor a;              /* saved activation output (sign for LeakyReLU, value for tanh) */
  GanTensor da;
  GanTensor da2;            /* optional */
  GanTensor dy;             /* out */
  int32_t act;
  float slope;
  float* dbias;             /* optional [c] (sum of dy over n,h,w) */
  int32_t accumulate;
  void* workspace;
  size_t workspace_bytes;
} GanActBwdDesc;
int gan_act_bwd(const GanActBwdDesc* d, gan_stream_t stream);

/* ---- losses --------------------------------------------------------------------------------- */
/* tf.keras.losses.BinaryCrossentropy(from_logits=True) against a constant target (base_gan.py:227-245).
 * x: fp32 logits [count]. loss_out[0] (+)= loss_scale * mean(bce). If dx != NULL:
 * dx[i*dx_pitch] = grad_scale * (sigmoid(x)-target)/count in `dtype`.  workspace >= 1024 floats (block partial
 * sums, added in a fixed order by a finalize launch). */
int gan_bce_logits(const float* x, int64_t count, float target, float loss_scale, int32_t loss_accumulate,
                   float* loss_out, float grad_scale, int32_t dtype, void* dx, int32_t dx_pitch, float* workspace,
                   const float* scale_state, gan_stream_t stream);
/* The three BCE terms of one Pix2Pix / PatchGAN step in one pass (generator_loss pix2pix.py:167-188 and
 * discriminator_loss base_gan.py:227-245 with the 0.5 of pix2pix.py:206): gan_loss = BCE(1, fake),
 * disc_loss = 0.5*(BCE(1, real) + BCE(0, fake)), gradients (optional, `dtype`, element stride `pitch`):
 * g_dfake = d gan_loss / d fake, d_dreal / d_dfake = d disc_loss / d real, fake.  If gen_total != NULL it receives
 * gan_loss + lambda * (*l1) (l1 = the already computed L1 term, pix2pix.py:184).  workspace >= 768 floats. */
int gan_patchgan_losses(const float* real_logits, const float* fake_logits, int64_t count, int32_t dtype, void* g_dfake,
                        void* d_dreal, void* d_dfake, int32_t pitch, float lambda, const float* l1, float* gen_total,
                        float* gan_loss, float* disc_loss, float* workspace, const float* scale_state, gan_stream_t stream);
/* tf.reduce_mean(tf.abs(a - b)) (pix2pix.py:181, cycle_gan.py:167,176). loss_out (+)= loss_scale*mean.
 * da (optional, dtype, own pitch) = grad_scale * sign(a-b)/count. workspace >= 4096 floats. */
int gan_l1(int32_t dtype, const GanTensor* a, const GanTensor* b, float loss_scale, int32_t loss_accumulate,
           float* loss_out, float grad_scale, const GanTensor* da, float* workspace, const float* scale_state,
           gan_stream_t stream);

/* out[k] = a[k] + b[k] + c[k], n <= 64 device scalars: total_gen_g_loss = gen_g_loss + total_cycle_loss + identity_loss
 * (cycle_gan.py:243-244) without leaving the captured step. */
int gan_sum3(const float* a, const float* b, const float* c, float* out, int32_t n, gan_stream_t stream);

/* ---- optimiser ------------------------------------------------------------------------------ */
/* tf.keras.optimizers.Adam (base_gan.py:247-252), TF form: lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
 * m += (g-m)(1-b1); v += (g*g-v)(1-b2); p -= lr_t*m/(sqrt(v)+eps).  `step` is a device counter:
 * gan_adam_begin increments it and writes lr_t, so a captured graph advances correctly on replay. */
int gan_adam_begin(int32_t* step, float* lr_t, float lr, float beta1, float beta2, const float* scale_state, gan_stream_t stream);
/* grad: fp32 [count], or with grad_bf16 != 0 the bfloat16 wire buffer of the data-parallel exchange (gan_grad_pack, summed
 * over ranks): Adam then reads the exchanged gradient where it landed, grad_scale = 1/world (8-byte aligned). */
int gan_adam_tf(float* param, float* m, float* v, const void* grad, int64_t count, const float* lr_t,
                float beta1, float beta2, float eps, float grad_scale, const float* scale_state, int32_t grad_bf16,
                gan_stream_t stream);

/* Dynamic loss scaling for the fp16 path (GAN_F16; the reference trains in fp32 and has none - this is what
 * tf.keras.mixed_precision.LossScaleOptimizer would add around base_gan.py:247-252).  scale_state: 4 floats on the
 * device {scale, 1/scale, finite steps in a row, this step's gradients are non-finite}.  Every loss entry point above
 * multiplies the GRADIENTS it writes by scale (reported losses stay unscaled); every Adam entry point multiplies the
 * gradients it reads by 1/scale and does nothing - gan_adam_begin included - while the non-finite flag is set.
 * Per step: backward -> gan_grads_check on each gradient buffer -> Adam -> gan_loss_scale_update (halves the scale after
 * a non-finite step, doubles it after growth_interval finite ones, clears the flag).  scale_state == NULL everywhere:
 * no scaling (fp32 / bf16 paths). */
int gan_grads_check(const float* grad, int64_t count, float* scale_state, gan_stream_t stream);
int gan_loss_scale_update(float* scale_state, int32_t growth_interval, float max_scale, gan_stream_t stream);

/* ---- misc ----------------------------------------------------------------------------------- */
/* Bernoulli(0.5) keep-mask from a counter hash of (seed, *step, stream_id, index). */
int gan_dropout_mask(uint8_t* mask, int64_t count, uint64_t seed, const int32_t* step, uint32_t stream_id,
                     gan_stream_t stream);
/* the same for n <= 4 masks (the three Dropout layers of a generator call) in a single launch; host arrays.  draws: optional
 * device int32[2], zero-initialised by the caller: a per-call-site launch counter mixed into the hash and advanced by the
 * launch itself, so that successive calls draw new masks while *step stands still (training=True forward passes of the
 * validation loop, pix2pix.py:228 / :338: Keras draws a fresh mask per call).  NULL = the single-mask hash above. */
int gan_dropout_mask_multi(int32_t n, uint8_t* const* masks, const int64_t* counts, uint64_t seed, const int32_t* step,
                           const uint32_t* stream_ids, int32_t* draws, gan_stream_t stream);
/* dst(dtype, pitch view) <- src fp32 dense [n,h,w,c]  /  dst fp32 dense <- src(dtype, pitch view) */
int gan_pack(int32_t dtype, const float* src, const GanTensor* dst, gan_stream_t stream);
/* n <= 4 (source, destination) pairs of one shape in a single launch (host arrays, read at call time) */
int gan_pack_multi(int32_t dtype, int32_t n, const float* const* srcs, const GanTensor* dsts, gan_stream_t stream);
int gan_unpack(int32_t dtype, const GanTensor* src, float* dst, gan_stream_t stream);
/* typed view -> typed view (same n,h,w,c; pitches / channel offsets may differ): places the generator
 * output into the discriminator's concat input (base_gan.py:139) and routes its gradient back. */
int gan_copy_view(int32_t dtype, const GanTensor* src, const GanTensor* dst, gan_stream_t stream);
/* bias gradient: dbias[c] (+)= sum over n,h,w of dy[...,c]; dy.c 8-padded, dbias has dy.c entries.
 * workspace >= gan_norm_workspace_bytes(1, dy.c, n*h*w). */
int gan_bias_grad(int32_t dtype, const GanTensor* dy, float* dbias, int32_t accumulate, void* workspace,
                  size_t workspace_bytes, gan_stream_t stream);
/* Wire format of the data-parallel gradient exchange (no counterpart in the reference, which is single-device:
 * base_gan.py:18-19 only prints the GPU count; semantics in SURVEY.md section 8e): fp32 gradients -> bf16 wire buffer
 * before the RCCL all-reduce, and back (times `scale` = 1/world) before Adam.  count % 8 == 0, 16-byte aligned. */
int gan_grad_pack(const float* src, void* dst_bf16, int64_t count, gan_stream_t stream);
int gan_grad_unpack(const void* src_bf16, float* dst, int64_t count, float scale, gan_stream_t stream);
/* Host utility (no GPU): CRC-32C of TensorFlow's TensorBundle checkpoint files (tf.train.Checkpoint /
 * CheckpointManager, pix2pix.py:400-403,419-420; cycle_gan.py:437-444,460-461).  crc = 0 to start; chainable. */
uint32_t gan_crc32c(uint32_t crc, const void* data, size_t n);
const char* gan_version(void);

/* ---- planner options -------------------------------------------------------------------------- */
/* The launch planners' tunable constants.  The library reads NO environment variable: these calls are the only way to
 * change them, and a change applies to the entry-point calls that follow it (each call plans for itself).  Unknown key:
 * GAN_E_ARG.  Keys (default): conv.big_tiles (1), conv.q128 (55), conv.q256n (80), conv.big_min_blocks (64),
 * conv.tall64 (1), conv.pingpong (1), conv.lean_epilogue (1), conv.tap_share (7: bit 0 = 256x128 tiles, bit 1 = 256x256 tiles on the tap-shared kernel, bit 2 = its table-driven form on the 256x128 tiles), conv.parity_patch (1), conv.parity_patch_max_n (64),
 * conv.parity_patch_min_blocks (192), conv.split_target (256), conv.split_target_skinny (1024),
 * conv.split_target_big (256), conv.split_target_256 (128), conv.split_min_ktiles (4), conv.split_max (64), conv.bwd_fuse_tile (1: the fused backward
 * epilogue rides on every tile epilogue; 0 never, 2 not on 64-column tiles, 3 on 64-column tiles only), conv.thin (7: bit 0
 * streaming kernels for the <= 8-channel layers, bit 1 thin-N, bit 2 thin-K), conv.norm_fuse (1), conv.stack (0: the callers in gan_amd/ merge eligible runs into layer stacks only when set), conv.stack_blocks (256), conv.thin_fused (1), wgrad.tile256 (0), wgrad.pingpong (1), wgrad.row_table (1),
 * wgrad.pingpong_min_rows (0 = automatic), wgrad.pingpong_128 (1), wgrad.pingpong_min_gflop (30),
 * wgrad.split_target (512), wgrad.fold_split_target (512), wgrad.reduce_adam (1),
 * wgrad.reduce_adam_min_params (1048576: smaller kernels keep the flat slab reduce and the caller's multi-tensor Adam pass),
 * conv.reduce_stats_rg (16: the slab reduce of a split-K launch that also emits the statistics partials walks up to this many row groups
 * per workgroup, one chunk each - keeps the chunk count of a 4,096-row 512-channel layer under ~512 so that the fused path is taken),
 * conv.own_max_rows (16) / conv.own_max_kb (192): a norm_fuse layer of at most this many GEMM rows per parity (<= 64; 0: never) whose
 * live taps x (8 weight rows + its rows) stay under this many KB per workgroup runs on the column-owner kernel (csrc/conv_own.hip)
 * instead of split-K GEMM + finishing slab reduce - same products, another fixed summation order,
 * wgrad.dead_taps (1: a fused-Adam wgrad launch skips the taps that never meet the map at its shape - 12 of 16 for a 2x2 -> 1x1
 * layer - where their moments are zero: the update is the identity there), diag.launch_log (0: see gan_launch_log below). */
int gan_set_option(const char* key, int32_t value);
int gan_get_option(const char* key, int32_t* value);
/* Diagnostic launch log (profiling tools only; no counterpart in the reference, whose only timing is time.time() per epoch,
 * pix2pix.py:261,319).  gan_set_option("diag.launch_log", 1) clears the log and starts recording the kernel symbol of EVERY
 * launch the entry points make, in enqueue order (0 stops).  gan_launch_log() copies the mangled names, newline-separated, into
 * buf (at most cap bytes, NUL-terminated) and returns the bytes the whole log needs.  tools/class_profile.py joins it with a
 * rocprofv3 --kernel-trace of the replayed step: per-launch class label + algorithmic FLOPs beside the measured duration. */
size_t gan_launch_log(char* buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
