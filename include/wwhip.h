/*
 * wwhip.h -- C-ABI of libwwhip.so: the MI355X (gfx950) hot path of the wakeword
 * training inner loop.
 *
 * The reference (sarpel/wakeword_trainer_home) is pure Python and has no FFI layer
 * (SURVEY.md F2); this ABI is what a maintainer binds with ctypes underneath the
 * reference's own Python interfaces (INTEGRATION.md shows the stub).  Every entry
 * point cites the reference interface whose device work it replaces.
 *
 * Conventions
 *   - extern "C"; every function returns int: 0 = ok, <0 = WW_E_* ; the message for
 *     the calling thread's last failure is ww_last_error().
 *   - All tensor arguments are DEVICE pointers owned by the caller (PyTorch-allocated);
 *     the library never frees or retains them.  Scratch is caller-provided.
 *   - Every launch goes to the hipStream_t passed as `stream` (void*; NULL = default
 *     stream).  No entry point synchronises the device or allocates device memory,
 *     except ww_ctx_create (uploads constant tables once).
 *   - Activations inside the conv stack are channels-last  [B][H][W][64] -- the memory format the
 *     reference trains in (src/training/trainer.py:71,165) -- stored as fp32 (WW_ACT_F32, the parity
 *     mode) or bf16 (WW_ACT_BF16; the counterpart of the reference's optimizer.mixed_precision switch,
 *     src/training/trainer.py:92,172).  Arithmetic, statistics and parameters are always fp32.
 *   - A ww_ctx is used by one host thread at a time (the training thread).
 */
#ifndef WWHIP_H
#define WWHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WW_ABI_VERSION 14

#define WW_OK 0
#define WW_E_INVALID (-1)     /* bad argument (shape, null pointer, unsupported size) */
#define WW_E_HIP (-2)         /* a HIP runtime call failed */
#define WW_E_WORKSPACE (-3)   /* caller workspace too small */
#define WW_E_UNSUPPORTED (-4) /* valid request outside what the kernels implement */

typedef struct ww_ctx ww_ctx;
typedef void *ww_stream_t; /* hipStream_t */

int ww_abi_version(void);
const char *ww_last_error(void);
int ww_ctx_create(int device, ww_ctx **out);
int ww_ctx_destroy(ww_ctx *ctx);

/* ------------------------------------------------------------------ device-resident step control (HIP graph replay)
 * A captured HIP graph bakes every by-value launch argument, but three things change from one training step to the next:
 * the Philox step of SpecAugment / audio augmentation / dropout, the optimizer's learning rate (schedulers,
 * src/training/optimizer_factory.py:281-333) and its step_state slot.  While a control block is BOUND to the ctx, every
 * entry point reads them from this 32-byte device struct at kernel RUN time:
 *   Philox step  = the call's `step` argument (now an OFFSET: 0 = the batch being trained, 1 = the batch whose input stage
 *                  runs one step ahead) + ctl->step;
 *   ww_clip_optim_step: lr = ctl->lr (cfg->lr ignored), parity = ctl->parity (argument ignored), and the step record goes
 *                  to stats_host_alt instead of stats_host when that parity is 1 (two pinned buffers, so the host may
 *                  read step k's record while step k+1 is running).
 * ww_step_ctl_advance is the one-thread kernel that opens a step: step += 1, parity ^= 1 (the first node of the captured
 * graph).  The host owns the struct's memory and writes lr / the initial step through ordinary copies between replays.
 * The reference has no graph capture; BASELINE config 5 asks for a "hipGraph-captured step".       */
typedef struct {
    uint64_t step;      /* launched-step counter */
    float lr;
    int32_t parity;     /* 0 | 1 */
    float loss_scale;   /* fp16 storage mode: the dynamic loss scale (ww_ce2_loss_fwd_bwd multiplies dlogits by it) */
    int32_t growth_tracker;
    int32_t reserved[2];
} ww_step_ctl;
int ww_ctx_bind_step_ctl(ww_ctx *ctx, const ww_step_ctl *ctl_dev /* device memory, 8-byte aligned; NULL unbinds */);
int ww_step_ctl_advance(ww_ctx *ctx, ww_stream_t stream);

/* ------------------------------------------------------------------ features
 * Replaces FeatureExtractor.__call__ (src/data/feature_extraction.py -- ABSENT from the
 * reference snapshot; contract from src/evaluation/evaluator.py:86-94,122-128 and the
 * shape law of src/export/onnx_exporter.py:316-320: (1, n_feat, N//hop + 1)), batched.
 * Spec: DESIGN.md "Feature spec" (periodic Hann, center/reflect, |rFFT|^2, HTK mel,
 * log(mel+eps); MFCC = orthonormal DCT-II).  n_mfcc == 0 selects log-mel.            */
typedef struct {
    int32_t sample_rate; /* src/config/defaults.py:15  (16000) */
    int32_t n_fft;       /* :18 (1024: the size the fused kernel is built for; any other power of two in [64, 4096]
                            takes a general radix-2 kernel -- validator.py:129 accepts 256 ... 4096) */
    int32_t hop;         /* :19 (160) */
    int32_t n_mels;      /* :20 (128; BASELINE config 2 uses 40); <= 128 */
    int32_t n_mfcc;      /* :17 ; 0 = log-mel output, else <= n_mels */
    float f_min;
    float f_max;   /* <= 0 -> sample_rate/2 */
    float log_eps; /* 1e-6 */
} ww_feat_cfg;

/* Replaces SpecAugment.__call__ (src/data/augmentation.py -- ABSENT; ctor pinned by
 * tests/test_training_pipeline.py:252-257, probabilities by src/config/defaults.py:90-91).
 * Integer index law: DESIGN.md "SpecAugment spec" / oracle/specaugment.py.            */
typedef struct {
    int32_t freq_mask_param;
    int32_t time_mask_param;
    int32_t n_freq_masks; /* n_freq_masks + n_time_masks <= 16 */
    int32_t n_time_masks;
    float freq_mask_prob;
    float time_mask_prob;
} ww_specaug_cfg;

#define WW_WAVE_F32 0
#define WW_WAVE_I16 1

int ww_feat_num_frames(int n_samples, int hop);

/* Host only (no GPU, no context): the HTK mel filterbank ww_logmel_fwd uses for cfg, as the kernels read it.  Replaces what
 * the reference gets from torchaudio's MelSpectrogram inside FeatureExtractor (src/data/feature_extraction.py is absent from
 * the snapshot; called from src/training/trainer.py:165-170 through the data pipeline).
 *   start, len (n_mels ints each): first spectrum bin / number of bins of each triangular band
 *   w (w_cap floats, nullable): the bands' weights back to back, *n_w of them
 *   melq_tab (WW_MELQ_TAB ints, nullable), melq_w (melq_cap floats, nullable), *n_melq_w (nullable): for n_fft 1024, the
 *   same weights in the form the v_mfma_f32_4x4x1 band sums of the STFT kernel read: [0] passes P, [1] 4-band quads NQ,
 *   [2+2p] steps of pass p, [3+2p] offset of its weights, [10+32p+2b] first bin of block b, [11+32p+2b] its unit number,
 *   [138+q] first unit of quad q; weight of (pass p, step t, lane) at melq_w[offset_p + 64 t + lane] = 1/4 of the weight of
 *   band 4*quad + lane%4 at bin first_bin(block lane/4) + t.                                                              */
#define WW_MELQ_TAB 172
int ww_feat_mel_tables(const ww_feat_cfg *cfg, int32_t *start, int32_t *len, float *w, int w_cap, int32_t *n_w,
                       int32_t *melq_tab, float *melq_w, int melq_cap, int32_t *n_melq_w);

/* wave (B,N) f32|i16  ->  out (B,1,n_feat,T) f32, optionally SpecAugment-masked in the
 * same pass.  mask_idx (nullable): int32 (B, n_f+n_t, 2) rows (start,width).           */
int ww_logmel_fwd(ww_ctx *ctx, const void *wave, int wave_dtype, int B, int N, const ww_feat_cfg *cfg,
                  float *out, const ww_specaug_cfg *sa, uint64_t seed, uint64_t step,
                  uint64_t sample_offset, int32_t *mask_idx, ww_stream_t stream);

/* How many persistent workgroups the log-mel kernel is launched with from now on: 0 (default) = one full residency round
 * of the device -- the front end running alone (feature extraction, validation); n > 0 = at most n -- a front end that
 * runs on a side stream BESIDE a training step, where its resident workgroups take registers / LDS from the conv kernels
 * (Trainer: one per CU).  A launch parameter only: results are identical for every n.            */
int ww_ctx_set_logmel_workgroups(ww_ctx *ctx, int n);

/* In-place SpecAugment on ready-made features x (B,1,F,T).                             */
int ww_specaug_apply(ww_ctx *ctx, float *x, int B, int F, int T, const ww_specaug_cfg *sa,
                     uint64_t seed, uint64_t step, uint64_t sample_offset, int32_t *mask_idx,
                     ww_stream_t stream);

/* ------------------------------------------------------------------ waveform augmentation
 * Replaces the RIR / background-noise part of AudioAugmentation.__call__ (src/data/augmentation.py -- ABSENT; ctor
 * tests/test_training_pipeline.py:230-236, knobs src/config/defaults.py:81-87).  Law: DESIGN.md "Audio augmentation
 * spec" / oracle/audio_augment.py.  rirs (R,L) f32, L <= 8192; noises (K,Nn) f32, Nn >= N; either bank may be absent
 * (NULL, 0).  choice_out (nullable): int32 (B,4) = rir index|-1, noise index|-1, noise offset, float bits of snr_db.
 * wave_out must not alias wave_in.                                                                                   */
typedef struct {
    float rir_prob;    /* defaults.py:87 */
    float noise_prob;  /* background_noise_prob, defaults.py:82 */
    float snr_min_db;  /* noise_snr_min, :83 */
    float snr_max_db;  /* noise_snr_max, :84 */
} ww_audio_aug_cfg;
size_t ww_audio_augment_scratch_bytes(int B, int N);
/* Optional, once per RIR bank: spectra for the FFT (overlap-save, 16384-point) form of the convolution, which
 * ww_audio_augment uses when `rir_spectra` is non-NULL; NULL selects the direct time-domain form (short RIRs). */
size_t ww_audio_rir_spectra_bytes(int R);
int ww_audio_rir_spectra(ww_ctx *ctx, const float *rirs, int R, int L, void *spectra, size_t spectra_bytes,
                         ww_stream_t stream);
int ww_audio_augment(ww_ctx *ctx, const float *wave_in, float *wave_out, int B, int N, const float *rirs, int R, int L,
                     const void *rir_spectra, const float *noises, int K, int Nn, const ww_audio_aug_cfg *cfg, uint64_t seed, uint64_t step,
                     uint64_t sample_offset, int32_t *choice_out, void *scratch, size_t scratch_bytes,
                     ww_stream_t stream);

/* ------------------------------------------------------------------ conv stack layers
 * Replace nn.Conv2d / nn.BatchNorm2d(train) / nn.ReLU as the reference composes them
 * (stem form: src/models/architectures.py:99-102; depthwise-separable blocks are the
 * torchvision MobileNetV3 building block the reference imports at :91-95).
 *
 * A "layer" here = conv producing the PRE-BatchNorm tensor y, plus the per-channel batch
 * statistics of y.  BatchNorm+ReLU of a layer is applied by its CONSUMER while loading
 * (scale/shift vector `ss` = [scale(64) | shift(64)]), so normalised activations never
 * touch HBM.  `mr` = [mean(64) | rstd(64)] is kept for backward.                        */
typedef struct {
    const float *gamma;  /* (64) BatchNorm2d.weight */
    const float *beta;   /* (64) BatchNorm2d.bias */
    float *running_mean; /* (64) updated when training != 0 */
    float *running_var;  /* (64) */
    float momentum;      /* 0.1 */
    float eps;           /* 1e-5 */
    int32_t training;    /* 1: batch statistics (+running update); 0: running statistics */
} ww_bn_t;

#define WW_C 64              /* channel width of the conv stack */
#define WW_ACT_F32 0         /* storage type of the activation tensors y_l / g_l (void* arguments) */
#define WW_ACT_BF16 1
#define WW_ACT_F16 2         /* fp16 storage / matrix operands: the reference's own AMP type (fp16 autocast + GradScaler,
                                src/training/trainer.py:172,182-193); gradients carry the loss scale (ww_loss_scale) */
#define WW_MAX_PARTIALS 1024 /* rows of a reduction slab */
/* scratch for one layer call: partial-sum slabs */
size_t ww_layer_scratch_bytes(void);

/* x (B,Hin,Win) f32 (C=1)  ->  y (B,Ho,Wo,64), Ho=(Hin+1)/2, Wo=(Win+1)/2 ; 3x3 s2 p1 */
int ww_conv_stem_fwd(ww_ctx *ctx, int act_dtype, const float *x, const float *w, int B, int Hin, int Win, void *y,
                     const ww_bn_t *bn, float *ss_out, float *mr_out, void *scratch, ww_stream_t stream);
/* depthwise 3x3 p1 on relu(bn(y_in)) */
int ww_dwconv3x3_fwd(ww_ctx *ctx, int act_dtype, const void *y_in, const float *ss_in, const float *w, int B, int H,
                     int W, void *y, const ww_bn_t *bn, float *ss_out, float *mr_out, void *scratch,
                     ww_stream_t stream);
/* pointwise 1x1 (64->64) on relu(bn(y_in)); f32 MFMA */
int ww_pwconv1x1_fwd(ww_ctx *ctx, int act_dtype, const void *y_in, const float *ss_in, const float *w, int B, int H,
                     int W, void *y, const ww_bn_t *bn, float *ss_out, float *mr_out, void *scratch,
                     ww_stream_t stream);
/* AdaptiveAvgPool2d(1) of relu(bn(y)):  pool (B,3,64) = [sum relu(z) | sum_{z>0} yhat | count_{z>0}] */
int ww_gap_fwd(ww_ctx *ctx, int act_dtype, const void *y, const float *ss, const float *mr, int B, int H, int W,
               float *pool, ww_stream_t stream);
/* dropout(Philox) + nn.Linear(64,2) (classifier of cnn_small; reference head form
 * src/models/architectures.py:105-111).  pd (B,64) = dropped pooled vector (kept for bwd) */
int ww_head_fwd(ww_ctx *ctx, const float *pool, int B, int HW, const float *fc_w, const float *fc_b,
                float dropout_p, int training, uint64_t seed, uint64_t step, uint64_t sample_offset, float *pd,
                float *logits, ww_stream_t stream);

/* ---- backward.  `coef` (192) = per-channel [A | Bc | Cc] with  dy = A*dz + Bc*y + Cc
 * (BatchNorm backward folded into one fma chain); produced for the INPUT layer of each
 * call together with that layer's dgamma/dbeta.                                          */
int ww_head_bwd(ww_ctx *ctx, const float *dlogits, const float *pd, const float *pool, int B, int HW,
                const float *fc_w, float dropout_p, int training, uint64_t seed, uint64_t step,
                uint64_t sample_offset, const float *gamma_last, const float *mr_last, float *dfc_w,
                float *dfc_b, float *dpool, float *coef_last, float *dgamma_last, float *dbeta_last,
                ww_stream_t stream);
/* g == NULL -> this is the last conv layer: dz = dpool[b][c] * [z>0] (dpool carries 1/HW).
 * WW_ACT_BF16: y_out is not read -- the kernel recomputes it from y_in with the forward's MFMA chain and rounding
 * (bit-identical to the stored tensor); it must still be passed (fp32 mode reads it).  B*H*W < 2^31.            */
int ww_pwconv1x1_bwd(ww_ctx *ctx, int act_dtype, const void *g, const float *dpool, const void *y_out,
                     const float *ss_out, const float *coef, const void *y_in, const float *ss_in, const float *mr_in,
                     const float *gamma_in, const float *w, int B, int H, int W, void *g_in, float *dw,
                     float *coef_in, float *dgamma_in, float *dbeta_in, void *scratch, ww_stream_t stream);
int ww_dwconv3x3_bwd(ww_ctx *ctx, int act_dtype, const void *g, const void *y_out, const float *coef, const void *y_in,
                     const float *ss_in, const float *mr_in, const float *gamma_in, const float *w, int B,
                     int H, int W, void *g_in, float *dw, float *coef_in, float *dgamma_in,
                     float *dbeta_in, void *scratch, ww_stream_t stream);
int ww_conv_stem_bwd(ww_ctx *ctx, int act_dtype, const void *g, const void *y_out, const float *coef, const float *x,
                     int B, int Hin, int Win, float *dw, void *scratch, ww_stream_t stream);

/* ------------------------------------------------------------------ whole model
 * cnn_small (added to create_model, src/models/architectures.py:437):
 * stem + 4 x (dw,pw) + GAP + dropout + Linear(64,2).  params/grads: arrays of
 * WW_CNN_SMALL_NPTR device pointers in state_dict order:
 *   0 stem.conv.weight  1 stem.bn.weight  2 stem.bn.bias  3 stem.bn.running_mean  4 stem.bn.running_var
 *   5+10i .. : blocks.i.dw.weight, dw_bn.{weight,bias,running_mean,running_var},
 *              blocks.i.pw.weight, pw_bn.{weight,bias,running_mean,running_var}      (i = 0..3)
 *   45 classifier.weight   46 classifier.bias
 * (grads: running_* slots are ignored).                                                  */
#define WW_CNN_SMALL_NPTR 47
size_t ww_cnn_small_workspace_bytes(int B, int F, int T, int act_dtype);
int ww_cnn_small_fwd(ww_ctx *ctx, int act_dtype, void *const *params, const float *x, int B, int F, int T, int training,
                     float bn_momentum, float bn_eps, float dropout_p, uint64_t seed, uint64_t step,
                     uint64_t sample_offset, void *ws, size_t ws_bytes, float *logits, ww_stream_t stream);
/* must follow a training-mode ww_cnn_small_fwd on the same ws/x.  part: WW_BWD_ALL, or the backward in two calls --
 * WW_BWD_LATE (classifier + blocks 3, 2: gradient slots 25..46) then WW_BWD_EARLY (blocks 1, 0 + stem: slots 0..24) -- so a
 * data-parallel caller can start the all-reduce of the late layers' gradients while the early layers' backward runs
 * (the reference is single-GPU, README.md:253-254). */
#define WW_BWD_ALL 0
#define WW_BWD_LATE 1
#define WW_BWD_EARLY 2
int ww_cnn_small_bwd(ww_ctx *ctx, int act_dtype, void *const *params, void *const *grads, const float *x, const float *dlogits,
                     int B, int F, int T, float dropout_p, uint64_t seed, uint64_t step, uint64_t sample_offset,
                     void *ws, size_t ws_bytes, int part, ww_stream_t stream);

/* ------------------------------------------------------------------ loss + step glue
 * Replaces LabelSmoothingCrossEntropy.forward (src/models/losses.py:66-98), the eps==0
 * nn.CrossEntropyLoss branch (:256) and FocalLoss.forward (:170-197) for C == 2, plus their
 * autograd, plus the per-batch accuracy / confusion counters of
 * src/training/trainer.py:196-200 + src/training/metrics.py:105-116.                     */
#define WW_LOSS_CE 0
#define WW_LOSS_FOCAL 1
typedef struct {
    float loss;
    float grad_norm; /* written by ww_grad_norm_clip */
    int32_t correct, tp, tn, fp, fn;
    int32_t nonfinite;  /* != 0 if the loss is not finite (trainer.py:177) */
    int32_t bad_target; /* != 0 if any target is outside [0,2) (losses.py:72) */
    int32_t count;      /* B */
    float found_inf;    /* 1.0 if this step must not update parameters (non-finite loss, bad target, or -- after
                           ww_grad_norm_clip -- non-finite gradient norm), else 0.0.  Layout-compatible with the
                           `found_inf` tensor of PyTorch's fused optimizers, so the reference's "skip the batch"
                           (trainer.py:177-179) needs no host round trip before optimizer.step(). */
    int32_t reserved;
} ww_step_stats;
/* Dynamic loss scaling of the fp16 storage mode (WW_ACT_F16) -- torch.amp.GradScaler's state and update rule
 * (create_grad_scaler, src/training/optimizer_factory.py:403-420; scale -> unscale_ -> step -> update,
 * src/training/trainer.py:182-193) kept and advanced on the DEVICE: ww_ce2_loss_fwd_bwd multiplies dL/dlogits by
 * scale[slot], every gradient down to the flat bucket then carries that factor, ww_clip_optim_step divides it out before the
 * norm / clip / update, and writes scale[slot^1], growth_tracker[slot^1]: non-finite gradients -> scale * backoff_factor and
 * the step is skipped; an applied step -> tracker + 1, and scale * growth_factor once it reaches growth_interval; a batch
 * skipped for its loss or targets leaves both unchanged.  slot alternates like ww_clip_optim_step's parity (it IS that parity).
 * GradScaler's defaults: scale 65536, growth 2, backoff 0.5, interval 2000.                                            */
typedef struct {
    float scale[2];
    int32_t growth_tracker[2];
    float growth_factor, backoff_factor;
    int32_t growth_interval;
    int32_t reserved;
} ww_loss_scale;
/* found_inf_out (nullable): receives stats->found_inf as a float of its own -- a data-parallel caller points it at the
 * spare last element of its flat gradient bucket, so the all-reduce of the gradients also tells every rank that SOME rank
 * must skip this batch (ww_clip_optim_step's found_inf_extra reads it back).  loss_scale (nullable) / loss_scale_slot: see
 * ww_loss_scale; with a bound ww_step_ctl the slot is the block's parity.                                              */
int ww_ce2_loss_fwd_bwd(ww_ctx *ctx, const float *logits, const int64_t *targets, int B, int loss_kind,
                        float label_smoothing, float focal_alpha, float focal_gamma, float *loss_out,
                        float *dlogits, ww_step_stats *stats, float *found_inf_out, const ww_loss_scale *loss_scale,
                        int loss_scale_slot, ww_stream_t stream);
/* Replaces clip_gradients -> torch.nn.utils.clip_grad_norm_
 * (src/training/optimizer_factory.py:446-452) on one flat gradient bucket.  max_norm <= 0:
 * only the norm is computed.  norm_out (nullable) receives the pre-clip L2 norm.          */
int ww_grad_norm_clip(ww_ctx *ctx, float *flat_grads, size_t n, float max_norm, float *norm_out,
                      ww_step_stats *stats /* nullable: grad_norm and found_inf are updated */, ww_stream_t stream);
/* ------------------------------------------------------------------ dense layers on the matrix cores (K8)
 * nn.Linear (+ Hardswish + Dropout) forward/backward: the classifier the reference puts on MobileNetV3,
 *   Sequential(Linear(576,1024), Hardswish(), Dropout(p), Linear(1024,num_classes))   (src/models/architectures.py:105-111).
 * x (M,K), w (N,K) = nn.Linear.weight, bias (N) nullable, y/pre/dy (M,N), dx (M,K), dw (N,K), db (N); all fp32, row-major.
 * mode WW_ACT_F32: fp32 MFMA (v_mfma_f32_32x32x2_f32), parity mode; WW_ACT_BF16: operands rounded to bf16, fp32
 * accumulation (v_mfma_f32_32x32x16_bf16).  epi (nullable): activation, then dropout drawn from the Philox stream
 * ctr = (step, sample_offset + row, TAG_DROPOUT<<24 | col>>2), lane col&3 (dropout_p = 0 in eval).  `pre` receives the
 * pre-activation the backward of the activation needs.                                                              */
#define WW_LIN_NONE 0
#define WW_LIN_HARDSWISH 1
#define WW_LIN_RELU 2          /* squeeze-excitation fc1 */
#define WW_LIN_HARDSIGMOID 3   /* squeeze-excitation fc2 (scale_activation of torchvision's MobileNetV3) */
typedef struct {
    int32_t act;
    float dropout_p;
    uint64_t seed, step, sample_offset;
} ww_linear_epi;
int ww_linear_mfma_fwd(ww_ctx *ctx, int mode, const float *x, const float *w, const float *bias, int M, int K, int N,
                       const ww_linear_epi *epi, float *pre /* nullable */, float *y, ww_stream_t stream);
size_t ww_linear_mfma_bwd_scratch_bytes(int M, int K, int N);
int ww_linear_mfma_bwd(ww_ctx *ctx, int mode, const float *x, const float *w, const float *pre /* nullable if no act */,
                       const float *dy, int M, int K, int N, const ww_linear_epi *epi, float *dx /* nullable */, float *dw,
                       float *db /* nullable */, void *scratch, size_t scratch_bytes, ww_stream_t stream);

/* The GEMM core for callers that keep operands in 16 bits IN HBM: C (M,N) = A (M,K) . B (N,K)^T, A / B bf16 (WW_ACT_BF16) or fp16
 * (WW_ACT_F16) row-major and 16-byte aligned, fp32 accumulation, C fp32 (c_f32 != 0) or the operand type.  K must be a
 * multiple of 64; M and N are free.  128 x 128 tiles, LDS-DMA operand staging.  B is in nn.Linear.weight
 * orientation, so y = x W^T needs no transpose; dx = dy W and dW = dy^T x take transposed copies of their right operand.   */
int ww_gemm16_nt(ww_ctx *ctx, int dtype, const void *A, const void *B, void *C, int c_f32, long M, long N, long K,
                 ww_stream_t stream);

/* ------------------------------------------------------------------ generic channels-last layers
 * Building blocks of bodies with varying channel counts (torchvision's mobilenet_v3_small, as MobileNetV3Wakeword
 * instantiates it: src/models/architectures.py:91-102).  Activations are fp32 row-major (M = B*H*W, C) matrices, so 1x1
 * convolutions and the squeeze-excitation FCs are ww_linear_mfma_* calls; these entry points cover the rest.
 * scratch: ww_nhwc_scratch_bytes(C) bytes.  act: WW_LIN_*.                                                         */
size_t ww_nhwc_scratch_bytes(int C);
/* BatchNorm2d (+activation): y = act(bn(x)); ss (2C) = scale|shift and mr (2C) = mean|rstd are kept for the backward.   */
int ww_bn_act_fwd(ww_ctx *ctx, const float *x, long M, int C, const ww_bn_t *bn, int act, float *y, float *ss, float *mr,
                  void *scratch, ww_stream_t stream);
int ww_bn_act_bwd(ww_ctx *ctx, const float *x, const float *da, long M, int C, const float *ss, const float *mr, int act,
                  int training, float *dx, float *dgamma, float *dbeta, void *scratch, ww_stream_t stream);
/* Conv2dNormActivation in TRAINING mode as the producer + one pass (torchvision's Conv2dNormActivation, the unit MobileNetV3 is
 * built from -- src/models/architectures.py:91-102): the convolution's own kernel leaves the BatchNorm statistics partials (the
 * GEMM epilogue's column sums per row tile / the LDS depthwise kernel's per image group), and the BatchNorm(+activation) apply
 * pass finishes them itself: 2 launches instead of conv + statistics + finish + apply (very tall layers keep the finish launch).
 * y = pre-BatchNorm output (kept for the backward), a = act(bn(y)); ss / mr as ww_bn_act_fwd.  scratch: ww_nhwc_scratch_bytes(C_out).
 * ww_conv1x1_bn_act_fwd: x (M,K) rows = pixels (or 3x3 patches of the stem), w (N,K), matrix mode as ww_linear_mfma_fwd.
 * ww_dwconv_bn_act_fwd: shapes the LDS kernel does not take (more than 128 input pixels per image, C % 4 != 0) and eval mode
 * run ww_dwconv_nhwc_fwd + ww_bn_act_fwd inside.                                                                      */
int ww_conv1x1_bn_act_fwd(ww_ctx *ctx, int mode, const float *x, const float *w, int M, int K, int N, const ww_bn_t *bn, int act,
                          const float *residual /* nullable (M,N): a = act(bn(y)) + residual, the block's skip connection */,
                          float *y, float *a, float *ss, float *mr, void *scratch, size_t scratch_bytes, ww_stream_t stream);
int ww_dwconv_bn_act_fwd(ww_ctx *ctx, const float *x, const float *w, int B, int H, int W, int C, int k, int stride,
                         const ww_bn_t *bn, int act, float *y, float *a, float *ss, float *mr, void *scratch, ww_stream_t stream);
/* depthwise Conv2d(C, C, k, stride, padding=k/2, groups=C, bias=False): x (B,H,W,C), w (C,1,k,k), k in {3,5}, stride in {1,2} */
int ww_dwconv_nhwc_fwd(ww_ctx *ctx, const float *x, const float *w, int B, int H, int W, int C, int k, int stride, float *y,
                       ww_stream_t stream);
int ww_dwconv_nhwc_bwd(ww_ctx *ctx, const float *x, const float *w, const float *dy, int B, int H, int W, int C, int k,
                       int stride, float *dx /* nullable */, float *dw, void *scratch, ww_stream_t stream);
/* AdaptiveAvgPool2d(1) on (B,HW,C) -> (B,C); per-(b,c) scaling y = x*gate and its two gradients; dx = dy*gate + dpool/HW   */
int ww_pool_hw_fwd(ww_ctx *ctx, const float *x, int B, int HW, int C, float *s, ww_stream_t stream);
int ww_scale_bc_fwd(ww_ctx *ctx, const float *x, const float *gate, int B, int HW, int C, float *y, ww_stream_t stream);
int ww_scale_bc_bwd_gate(ww_ctx *ctx, const float *x, const float *dy, int B, int HW, int C, float *dgate, ww_stream_t stream);
int ww_scale_pool_bwd(ww_ctx *ctx, const float *dy /* nullable */, const float *gate, const float *dpool /* nullable */, int B,
                      int HW, int C, float *dx, ww_stream_t stream);
/* Squeeze-excitation as torchvision's mobilenet_v3_small builds it (the model src/models/architectures.py:91-102 instantiates):
 * y = x * hardsigmoid(W2 relu(W1 mean_hw(x) + b1) + b2), x / y (B,HW,C), w1 (Cs,C), w2 (C,Cs) in nn.Conv2d(.,.,1) layout.  ONE launch
 * forward (a workgroup owns whole images), TWO backward (dx + per-image pre-activation gradients, then the four parameter
 * gradients as fixed-order sums over the batch); fp32 arithmetic whatever the model's matrix mode.  s (B,C), pre1 (B,Cs),
 * pre2 (B,C) are kept for the backward.  C % 4 == 0, Cs % 4 == 0, C <= 1024, Cs <= 256, else WW_E_UNSUPPORTED (callers then compose the block
 * from ww_pool_hw_fwd / ww_linear_mfma_* / ww_scale_bc_*).                                                              */
size_t ww_se_bwd_scratch_bytes(int B, int C, int Cs);
int ww_se_fwd(ww_ctx *ctx, const float *x, int B, int HW, int C, int Cs, const float *w1, const float *b1, const float *w2,
              const float *b2, float *y, float *s, float *pre1, float *pre2, ww_stream_t stream);
int ww_se_bwd(ww_ctx *ctx, const float *x, const float *dy, const float *s, const float *pre1, const float *pre2, const float *w1,
              const float *w2, int B, int HW, int C, int Cs, float *dx, float *dw1, float *db1, float *dw2, float *db2,
              void *scratch, size_t scratch_bytes, ww_stream_t stream);
/* The one-channel stem of MobileNetV3Wakeword (src/models/architectures.py:94-101 replaces torchvision's first conv by
 * Conv2d(1, 16, 3, stride 2, padding 1, bias=False)) as a direct convolution: x (B,H,W), w (C,1,3,3), C % 4 == 0, C <= 64.
 * Forward in training mode = convolution (which leaves the BatchNorm statistics partials) + the finishing apply pass, as
 * ww_conv1x1_bn_act_fwd; ww_stem3x3s2_bwd_dw is its weight gradient (the input needs none).  scratch: ww_nhwc_scratch_bytes(C). */
int ww_stem3x3s2_bn_act_fwd(ww_ctx *ctx, const float *x, const float *w, int B, int H, int W, int C, const ww_bn_t *bn, int act,
                            float *y, float *a, float *ss, float *mr, void *scratch, ww_stream_t stream);
int ww_stem3x3s2_bwd_dw(ww_ctx *ctx, const float *x, const float *dy, int B, int H, int W, int C, float *dw, void *scratch,
                        ww_stream_t stream);
/* 3x3 stride-2 pad-1 patches of a one-channel image (B,H,W) -> (B*ceil(H/2)*ceil(W/2), 9): the stem conv becomes a GEMM */
int ww_im2col3x3s2(ww_ctx *ctx, const float *x, int B, int H, int W, float *cols, ww_stream_t stream);
int ww_add_f32(ww_ctx *ctx, const float *a, const float *b, size_t n, float *y, ww_stream_t stream);

/* ------------------------------------------------------------------ conv front-end of the CRNN
 * cnn_small's conv stack (stem + 4 depthwise-separable blocks) without GAP / classifier, followed by the mean over the
 * frequency axis of the last layer's activations: x (B,1,F,T) -> seq (B, ceil(T/2), 64) fp32, the input sequence of the
 * recurrent layers.  Same parameter / gradient pointer tables and workspace as ww_cnn_small_fwd/bwd (entries 45, 46 --
 * the classifier -- are not read and may be NULL).  The reference has no CRNN: the topology is this
 * build's, assembled from the reference's conv idiom and its GRUWakeword.                                          */
int ww_cnn_front_fwd(ww_ctx *ctx, int act_dtype, void *const *params, const float *x, int B, int F, int T, int training,
                     float bn_momentum, float bn_eps, void *ws, size_t ws_bytes, float *seq, ww_stream_t stream);
int ww_cnn_front_bwd(ww_ctx *ctx, int act_dtype, void *const *params, void *const *grads, const float *x, const float *dseq,
                     int B, int F, int T, void *ws, size_t ws_bytes, ww_stream_t stream);

/* ------------------------------------------------------------------ GRU layer, one direction
 * torch.nn.GRU's cell and parameter layout (gate order r|z|n; w_ih (3H,I), w_hh (3H,H), b_ih, b_hh (3H)) -- what the
 * reference's GRUWakeword wraps (src/models/architectures.py:228-235).  batch_first: x (B,T,I) with row stride ldx
 * between consecutive (b,t) rows, y (B,T,H) with row stride ldy (a bidirectional layer's two directions write the two
 * halves of one (B,T,2H) buffer: ldy = 2H, y offset by H for the reverse direction).  reverse != 0 runs t = T-1..0.
 * h0 nullable (zeros).  H = 128 only.  ws (ww_gru_workspace_bytes, 256-byte aligned) keeps the projections and the
 * gates between ww_gru_fwd and the ww_gru_bwd of the same (layer, direction).  ww_gru_bwd: dy (nullable) is the
 * gradient of y, dh_n (nullable) of the final hidden state; dx is written, or added to when accumulate_dx != 0.     */
size_t ww_gru_workspace_bytes(int B, int T, int I, int H);
/* Dropout on a (B,T,C) tensor of row-strided rows (nn.GRU's inter-layer dropout, the Dropout in front of GRUWakeword.fc,
 * architectures.py:232,239): out = keep ? x/(1-p) : 0 with keep from Philox ctr = (step, sample_offset + b,
 * TAG_DROPOUT<<24 | stream_id<<20 | t<<8 | c>>2), lane c&3.  Applying it to the gradient is its backward.          */
int ww_dropout_bt(ww_ctx *ctx, const float *x, long ldx, int B, int T, int C, float p, uint64_t seed, uint64_t step,
                  uint64_t sample_offset, int stream_id, float *out, long ldo, ww_stream_t stream);
int ww_gru_fwd(ww_ctx *ctx, int mode /* WW_ACT_F32 | WW_ACT_BF16: matrix type of the projection GEMMs */, const float *x, long ldx, const float *w_ih, const float *w_hh, const float *b_ih,
               const float *b_hh, const float *h0, int B, int T, int I, int H, int reverse, float *y, long ldy, float *h_n,
               void *ws, size_t ws_bytes, ww_stream_t stream);
int ww_gru_bwd(ww_ctx *ctx, int mode, const float *x, long ldx, const float *w_ih, const float *w_hh, const float *dy, long ldy,
               const float *dh_n, int B, int T, int I, int H, int reverse, void *ws, size_t ws_bytes, float *dx, long lddx,
               int accumulate_dx, float *dw_ih, float *dw_hh, float *db_ih, float *db_hh, float *dh0, ww_stream_t stream);

/* Both directions of a bidirectional layer (nn.GRU(bidirectional=True), architectures.py:228-235) with ONE recurrent launch per
 * pass: the persistent kernel runs direction 0 (t = 0..T-1) and direction 1 (t = T-1..0) as the two rows of its grid -- twice
 * the resident workgroups of a per-direction launch, no second stream, so a captured HIP graph keeps the concurrency that two
 * streams give an eager step.  y / dy: (B,T,2H) with row stride ldy >= 2H, direction d owns columns [d*H, (d+1)*H).  Each
 * direction has its own workspace (ww_gru_workspace_bytes, 256-byte aligned, kept from forward to backward).  Backward-only
 * members may be NULL in the forward call and vice versa; h0 / h_n / dh_n / dh0 are nullable.  dx = sum over both directions. */
typedef struct ww_gru_dir {
    const float *w_ih, *w_hh, *b_ih, *b_hh;      /* nn.GRU layout: (3H,I), (3H,H), (3H), (3H); gate order r|z|n */
    const float *h0;                             /* (B,H) initial state or NULL (zeros) */
    float *h_n;                                  /* (B,H) final state out, or NULL */
    void *ws;
    const float *dh_n;                           /* backward: gradient of h_n, or NULL */
    float *dw_ih, *dw_hh, *db_ih, *db_hh;        /* backward: parameter gradients (written) */
    float *dh0;                                  /* backward: gradient of h0 out, or NULL */
} ww_gru_dir;
int ww_gru_bidir_fwd(ww_ctx *ctx, int mode, const float *x, long ldx, const ww_gru_dir *dir /* [2] */, int B, int T, int I, int H,
                     float *y, long ldy, size_t ws_bytes, ww_stream_t stream);
int ww_gru_bidir_bwd(ww_ctx *ctx, int mode, const float *x, long ldx, const ww_gru_dir *dir /* [2] */, const float *dy, long ldy,
                     int B, int T, int I, int H, size_t ws_bytes, float *dx, long lddx, ww_stream_t stream);

/* Deferred partial sums.  The parameter-gradient kernels of the generic layers (ww_linear_mfma_bwd's split-K dW product,
 * ww_dwconv_nhwc_bwd, ww_stem3x3s2_bwd_dw) end in "sum the per-block partials into the gradient" -- a 4-5 us launch each that
 * nothing needs before the optimizer.  While ww_ctx_set_deferred_reduce(ctx, 1) is in force those calls QUEUE that last step
 * (their `scratch` argument then holds the partials and must stay untouched until the flush; dw / the gradient output is not
 * valid before it) and ww_deferred_reduce_flush runs everything queued as ONE launch (fixed order, double accumulation).
 * The autograd models flush at the end of backward (and before a mid-backward all-reduce).  Replaces nothing in the reference:
 * its weight gradients come out of cuDNN / cuBLAS calls (src/training/trainer.py:182).                                      */
int ww_ctx_set_deferred_reduce(ww_ctx *ctx, int on);
int ww_deferred_reduce_pending(ww_ctx *ctx);
int ww_deferred_reduce_flush(ww_ctx *ctx, ww_stream_t stream);
int ww_deferred_reduce_discard(ww_ctx *ctx);      /* forget what is queued (the backward pass that queued it did not complete) */

/* Fused clip + optimizer step on flat fp32 buckets.  Replaces, for one step, the reference's
 * clip_gradients(...) ; optimizer.step()  (src/training/trainer.py:185-193) with torch.optim's own update rules
 * (create_optimizer, src/training/optimizer_factory.py:165-199: Adam, AdamW, SGD with nesterov=True).
 * max_norm <= 0: norm only.  exp_avg doubles as SGD's momentum buffer; exp_avg_sq is unused for SGD (may be NULL).
 * step_state: int64[2] on the device, both 0 initially; call k reads slot[parity] and writes slot[parity^1]
 * (callers alternate parity 0,1,0,...).  When stats->found_inf != 0 (or the gradient norm is not finite) nothing is
 * updated and the step count does not advance -- the reference's "skip this batch" (trainer.py:177-179).
 * stats_host (nullable): PINNED host memory (hipHostMalloc / torch pin_memory) that receives a copy of *stats, written
 * by the kernel itself -- the step's single device->host record without a copy command in the stream.              */
enum { WW_OPT_ADAM = 0, WW_OPT_ADAMW = 1, WW_OPT_SGD = 2 };
typedef struct {
    int32_t kind;
    float lr, beta1, beta2, eps, weight_decay, momentum;
    float max_norm;    /* gradient_clip (src/config/defaults.py:48) */
} ww_optim_cfg;
int ww_clip_optim_step(ww_ctx *ctx, const ww_optim_cfg *cfg, float *flat_params, float *flat_grads, float *exp_avg,
                       float *exp_avg_sq, size_t n, int64_t *step_state, int parity, float *norm_out,
                       ww_step_stats *stats /* nullable */, ww_step_stats *stats_host /* nullable */,
                       ww_step_stats *stats_host_alt /* nullable; see ww_step_ctl */,
                       const float *found_inf_extra /* nullable: != 0 -> skip (another rank's verdict) */,
                       ww_loss_scale *loss_scale /* nullable: fp16 mode, slot = parity */, ww_stream_t stream);

/* ------------------------------------------------------------------ collectives: deliberately NOT in this ABI
 * SURVEY.md §8b sketched two more entry points, `ww_comm_init` (ctx, nccl_unique_id, rank, world) and `ww_allreduce_f32`
 * (ctx, buf, n, avg, comm_stream).
 * They are waived: the exchange step of this path is ONE averaged all-reduce of the flat fp32 gradient bucket (two
 * ranges of it, see WW_BWD_LATE/EARLY above), and the host side of the boundary is Python on PyTorch-ROCm, whose
 * torch.distributed "nccl" backend IS RCCL over xGMI -- a second communicator owned by this library would duplicate
 * rendezvous, stream ordering and error handling that the host already has, and would add nothing on the device side
 * (RCCL's kernels are the collective; there is no fused reduce+clip kernel to hide behind an entry point).  What the ABI
 * does provide for data parallelism: gradients in one contiguous bucket (`grads` table -> flat buffer), the backward in
 * two calls so the first range can be reduced under the second, the found_inf slot that travels with the gradients
 * (ww_ce2_loss_fwd_bwd / ww_clip_optim_step), and Philox streams addressed by GLOBAL sample index (sample_offset).
 * The reference itself is single-GPU (README.md:253-254).                                                            */

/* ------------------------------------------------------------------ measurement
 * Opt-in timing of kernel classes with hipEvents recorded on the launch stream around the
 * class's main kernel (bench.py's roofline leg; no reference counterpart -- the reference never
 * measures, src/ui/panel_training.py:69,484).  ww_prof_collect synchronises on the recorded events
 * and returns, per class, the summed milliseconds and the number of launches since the last call. */
int ww_prof_num_classes(void);
const char *ww_prof_class_name(int cls);
int ww_prof_enable(ww_ctx *ctx, uint32_t class_mask);
int ww_prof_collect(ww_ctx *ctx, float *ms_sum, int32_t *count);

/* floor(p * 2^32) clamped to [0, 2^32] -- the integer probability threshold of the specs  */
uint64_t ww_prob_threshold(double p);
/* Philox4x32-10, exported so host tests can pin the device RNG's law (oracle/philox.py)   */
void ww_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

#ifdef __cplusplus
}
#endif
#endif /* WWHIP_H */
