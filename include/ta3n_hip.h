/*
 * ta3n_hip.h - C ABI of libta3n_hip.so, the MI355X (gfx950) implementation of
 * TA3N's temporal-adversarial train step.
 *
 * The reference (cmhungsteve/TA3N) is pure Python/PyTorch and has no FFI; the
 * drop-in boundary is its Python surface (models.VideoModel.forward,
 * models.py:545-722; main.train, main.py:309-667).  This header is what a
 * Python/ctypes (or any other FFI) host binds to replace that path.  Every
 * entry point cites the reference code it replaces.  All device pointers are
 * plain fp32/int32 HIP device memory owned by the caller; `stream` is a
 * hipStream_t passed as void*.  No entry point allocates per call or
 * synchronises the device (graph-capture safe) except ta3n_plan_create /
 * ta3n_plan_destroy / ta3n_set_hyper_sync.
 *
 * Layout conventions (all row-major fp32):
 *   x       [B*T, D]   source videos first (rows [0, Bs*T)), then target
 *   params  flat buffer, layout given by ta3n_param_info (live params first)
 *   grads   same layout as params (only the live prefix is written)
 *   ws      workspace of ta3n_workspace_floats() floats; named regions are
 *           located with ta3n_ws_offset()
 * Return value: 0 on success, negative ta3n_status otherwise;
 * ta3n_last_error() gives a message for the calling thread.
 */
#ifndef TA3N_HIP_H
#define TA3N_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    TA3N_OK = 0,
    TA3N_ERR_INVALID = -1,     /* bad argument / unsupported configuration (ValueError in the reference) */
    TA3N_ERR_HIP = -2,         /* a HIP runtime call failed */
    TA3N_ERR_NOMEM = -3
} ta3n_status;

/* flags for ta3n_config.flags */
#define TA3N_FLAG_ADV_RELATION   (1u << 0)  /* place_adv[0]=='Y'  (opts.py:67, main.py:508-538) */
#define TA3N_FLAG_ADV_VIDEO      (1u << 1)  /* place_adv[1]=='Y' */
#define TA3N_FLAG_ADV_FRAME      (1u << 2)  /* place_adv[2]=='Y' */
#define TA3N_FLAG_ATTN_ENTROPY   (1u << 3)  /* add_loss_DA=='attentive_entropy' (main.py:559-562) */
#define TA3N_FLAG_TRANS_ATTN     (1u << 4)  /* use_attn=='TransAttn' (models.py:643-645) */
/* ens_DA=='MCD' (opts.py:49; models.py:276-279, 716-720): a second video classifier "fc_classifier_video_source_2" on the
 * same (dropped-out) video feature; its logits land in region "Y2", its logit gradient is read from "gY2".  The MCD loss
 * itself (main.py:447, 548-556) is assembled by the caller from the two forwards; unfused entry points only. */
#define TA3N_FLAG_MCD            (1u << 5)
/* The backward pass accepts a caller-written gradient at the pooled video feature (region "gV_ext", [B][256], the feature
 * the reference returns as feat[1], models.py:675) in addition to the logit gradients: entry point of the discrepancy losses
 * (dis_DA DAN / JAN, main.py:452-505).  Unfused entry points only; the region is zero after ta3n_init_workspace. */
#define TA3N_FLAG_FEATURE_GRADS  (1u << 6)
/* use_bn 'AdaBN' / 'AutoDIAL' (models.py:195-198, 490-543, 569-570): a BatchNorm1d per domain ("bn_shared_S", "bn_shared_T":
 * weight and bias are parameters of the plan) between the shared frame FC and its ReLU.  Train mode: batch statistics over
 * the domain's rows (region "bn_batch" [2 domains][3][F] = mean, biased variance, 1/sqrt(var + eps)), and - round 6 - the launch
 * itself moves the running averages in region "bn_run" [2][2][F] = mean, variance the way nn.BatchNorm1d does (momentum 0.1,
 * unbiased variance); eval mode (hyper.train == 0) normalises with that region (a caller may also write it: loading a checkpoint).
 * alpha = 1 (no source/target batch mixing: what the reference's own program always runs).  Unfused entry points AND - round 6 -
 * the fused ta3n_train_step / ta3n_train_steps sequences of both aggregations: two BatchNorm launches behind the shared-FC product and
 * in front of its weight gradient; their gradients' share of the clip norm sits in the last slots of region "sumsq". */
#define TA3N_FLAG_BN_SHARED      (1u << 7)
/* Arithmetic of BASELINE.json configs[1]: every contraction rounds its two operands to bf16 (round to nearest
 * even) and multiplies them on the bf16 MFMA with fp32 accumulation.  Parameters, optimiser state, gradients and
 * everything outside the contractions (biases, softmax, losses, update) stay fp32.  Off = fp32 MFMA (configs[2]). */
#define TA3N_FLAG_BF16_MFMA      (1u << 8)
/* With TA3N_FLAG_BF16_MFMA: the forward contractions of ta3n_train_step read their operands from bf16 copies
 * ("twins", inside ws) instead of rounding fp32 values on the fly - same arithmetic, half the operand bytes.  The
 * library keeps the twins of everything it writes itself (activations: the producing kernel; parameters:
 * ta3n_sgd_step*); whatever the CALLER writes - the input features, parameters loaded from a checkpoint - must be
 * followed by ta3n_refresh_bf16.  (A bf16 feature store can feed the input twin directly.) */
#define TA3N_FLAG_BF16_STORE     (1u << 9)
/* fp32-grade contractions on the bf16 matrix cores ("bf16x3"): every operand x is split in registers into hi = bf16(x) and
 * lo = bf16(x - hi) (both round-to-nearest-even), and a product a b is accumulated in fp32 as a_hi b_hi + a_hi b_lo + a_lo b_hi -
 * 16 mantissa bits per operand, a relative error of ~2^-16 per product instead of fp32's 2^-24 (the dropped a_lo b_lo term),
 * at 3/16 of the fp32 MFMA's cycles.  Stage images, parameters, gradients and everything outside the MFMA stay fp32.  It is
 * NOT IEEE fp32 multiplication; it is held to the same parity tests as the fp32 configuration (logits within 1e-3 of the
 * reference's CPU path, gradients rtol 2e-4).  Exclusive with TA3N_FLAG_BF16_MFMA.
 * With TA3N_FLAG_BF16_STORE ("pair twins"): the split is made ONCE by whoever produces the data - every twin region of ws gets a
 * second plane holding lo beside hi (regions "ws16_lo", "p16_lo", "x16_lo"; GEMM epilogues, the heads kernel, the optimiser,
 * ta3n_refresh_bf16 and the feature-store gathers write both planes; rows of a bf16 feature store are their own hi plane, lo = 0) -
 * and the launches of ta3n_train_step whose operands qualify stream both planes (the same bytes as the fp32 images) and multiply
 * without converting anything in the K loop; the others keep splitting fp32 operands in registers.  Same products, other
 * summation order inside the MFMAs.  ta3n_train_steps_fused_update is not built for it (ta3n_has_fused_update returns 0). */
#define TA3N_FLAG_F32_SPLIT      (1u << 10)

/* ta3n_config.aggregation */
#define TA3N_AGG_TRN_M    0   /* 'trn-m': multi-scale TRN - the TA3N path (TRNmodule.py:27-86) */
#define TA3N_AGG_AVGPOOL  1   /* 'avgpool' (TemPooling, models.py:246, 421-433) in the source-only configuration of
                               * BASELINE configs[0] (script_train_val.sh:103-119: use_target none, every DA option off):
                               * flags must carry no TA3N_FLAG_ADV_x / _ATTN_ENTROPY / _TRANS_ATTN bit.  The step is
                               * F1 -> mean over segments -> dropout -> classifier -> CE on the source rows -> backward;
                               * the discriminators (forwarded by the reference, feeding nothing) are not computed. */

typedef struct {
    int32_t batch_source;    /* Bs: rows of the source half (after the reference's zero padding, main.py:359-372) */
    int32_t batch_target;    /* Bt */
    int32_t num_segments;    /* T  (opts.py:12; train_segments, models.py:60) */
    int32_t feature_dim;     /* D  (models.py:125-126) */
    int32_t fc_dim;          /* F = min(fc_dim, feature_dim) (models.py:129) */
    int32_t num_bottleneck;  /* 256 for trn-m (models.py:223) */
    int32_t num_class;       /* C */
    uint32_t flags;          /* TA3N_FLAG_* */
    int32_t tile_config;     /* 0 = auto; otherwise WM*100+WN*10+WK (114, 118, 212, 122, 214, 124, 221, 222) for every GEMM phase;
                              * + 2000 / 3000: LDS stages of the bf16 kernels (default: 3 when every tile streams K >= 1024);
                              * + 6000 / 7000: three / four HALF stages (64 k each) of the bf16-twin kernel (222 only);
                              * + 10000 / 20000 / 30000: 2 x 1 / 1 x 2 / 2 x 2 32x32 blocks per wave (bf16-twin kernel: 128x64, 64x128,
                              * 128x128 tiles); 46221 / 56221: 192x128 / 256x128 (four waves, three half stages; launches whose A
                              * operands are K-contiguous, the builder's own choice elsewhere).  Codes a launch cannot use fall back. */
    int32_t phase_tiles[16]; /* per GEMM phase (in launch order) override of tile_config; 0 = use tile_config/auto */
    int32_t xcd_aware;       /* 0 = default (on), 1 = on, 2 = off: order tiles so panels sharing an operand sit on one XCD; 3 = on, with affinity groups (specs that share operand slabs stay on one XCD, back to back) */
    int32_t aggregation;     /* TA3N_AGG_*: frame aggregation (opts.py --frame_aggregation) */
    int32_t wgrads_late;     /* fused step, 0 (default): the TRN / frame-discriminator weight gradients share the launch of the
                              * gradient at the frame features; 1: that launch holds only the (critical-path) gradient at the frame
                              * features and the weight gradients nobody waits for ride in the LAST launch, beside the shared-FC
                              * weight gradient.  Measured slower at the headline shape (17.0 + 19.5 us against 22.1 + 9.3 us: a
                              * long-K tile alone on a compute unit is bound by the latency of its two LDS stages, not by the
                              * load path it no longer shares); kept for other shapes and A/B runs. */
    int32_t chain;           /* fused step, 1: consecutive GEMM dependency levels share ONE launch - forward: shared frame FC -> {frame
                              * discriminator hidden layer, TRN tuple GEMMs} -> relation-discriminator hidden layer; backward: gradient at the
                              * frame features + weight gradients -> shared-FC weight gradient - with tile-level hand-offs inside the launch
                              * (a producer tile publishes its stores and bumps a counter, a consumer tile polls the counters of exactly
                              * the tiles it reads; derived from the tiles' read / write spans).  5 launches per step instead of 8, the
                              * same tiles and arithmetic: bit-identical results.  0: one launch per level.
                              * EXPERIMENTS BUILD ONLY (round 5; measured 144-190 us against 112): the plan is built either way (and
                              * checked on the CPU), launching it on the default library returns TA3N_ERR_INVALID with a message naming
                              * the build flag -DTA3N_EXPERIMENTS=1.  The same holds for split_k and ta3n_train_steps_fused_update. */
    int32_t cost_model;      /* how the plan orders a launch's tiles over the 8 XCD queues: 0 = by the length of their K loops; 1 = by an
                              * estimate of their TIME (fixed per-tile overhead + K, weight-gradient tiles weighted up).  Ordering only:
                              * results are bit-identical. */
    int32_t split_k;         /* fused step, 2: every tile of the gradient at the frame features (the longest K loops of the step: up to
                              * 3 072 deep, and the launch lasts as long as one of them) is computed by TWO workgroups, each over half of
                              * the tile's K segments.  Each publishes its partial tile (write-through stores) and takes a ticket; the
                              * second to arrive adds the other's partial to its own and runs the epilogue.  a + b = b + a: the result
                              * does not depend on who arrives last; it differs from the unsplit tile by fp32 summation order.
                              * Bit mask since round 6: 2 = those tiles; 4 = every tile of the step's first launch (shared-FC product, one
                              * tile per compute unit), its single K segment halved (measured slower in both arithmetics,
                              * profiles/r06_split_k_shared_fc_ab.txt); 6 = both.
                              * 0: one workgroup per tile.  Experiments build only (measured 114.8 us against 113.6). */
    int32_t reserved[1];
} ta3n_config;

/* Per-step scalars; lives in device memory inside ws (region "hyper").  The host
 * fills a copy and uploads it with ta3n_set_hyper (async on `stream`). */
typedef struct {
    float beta[3];           /* [relation, video, frame] GradReverse weights (models.py:20-29, main.py:350-352) */
    float gamma;             /* attentive-entropy weight (main.py:562) */
    float lr;                /* main.py:83, 620-621 */
    float momentum;          /* opts.py:83 */
    float weight_decay;      /* opts.py:85 */
    float clip;              /* --clip_gradient (main.py:578-581); <= 0 disables */
    float p_drop_i;          /* dropout_i (models.py:131); used only when train != 0 */
    float p_drop_v;          /* dropout_v (models.py:132) */
    uint32_t seed_i;         /* dropout stream seeds for this step */
    uint32_t seed_v;
    float inv_n_cls;         /* 1 / (global number of labelled source videos)      (CE mean, main.py:446) */
    float inv_n_rel;         /* 1 / (global src+tgt videos * (T-1))                 (main.py:533) */
    float inv_n_vid;         /* 1 / (global src+tgt videos) */
    float inv_n_frm;         /* 1 / (global src+tgt videos * T) */
    float inv_n_ent;         /* 1 / (global src+tgt videos)                         (loss.py:24) */
    int32_t valid_source;    /* rows [valid_source, Bs) are the reference's dummy rows (main.py:359-372, 421-422) */
    int32_t valid_target;
    int32_t train;           /* nn.Module.training: dropout active */
    int32_t reverse;         /* VideoModel.forward(..., reverse=True) (models.py:682-684): GradReverse(mu) on the video feature */
    float mu;                /* ... its weight (the MCD step's second forward, main.py:550) */
    int32_t reserved[2];
} ta3n_hyper;

typedef struct ta3n_plan ta3n_plan;

/* ---- integer layer (host, bit-exact contract) -------------------------------- */

/* Number of frame tuples RelationModuleMultiScale.forward uses for T frames:
 * 1 + sum_{s=T-1..2} min(3, C(T,s))   (TRNmodule.py:32-41, 60, 68-71). */
int ta3n_num_relation_tuples(int num_frames);

/* The tuples themselves, scale T first, by combinatorial unranking (no C(T,s)
 * enumeration).  tuples is [n][T] padded with -1; scale_len[r] = tuple size,
 * scale_id[r] = index of the scale (0 = T-frame).  Replaces
 * TRNmodule.py:30-41, 84-86 + the idx computation at :71.  Returns n or <0. */
int ta3n_relation_table(int num_frames, int32_t *tuples, int32_t *scale_len, int32_t *scale_id);

/* TSNDataSet._get_test_indices (dataset.py:103-116), 1-based frame ids.
 * Returns TA3N_ERR_INVALID where the reference raises (no selectable frame). */
int ta3n_segment_indices(int num_frames, int num_segments, int new_length, int64_t *out);

/* TSNDataSet.__getitem__ for a whole batch on the device (dataset.py:118-144 with the test-mode sampler
 * dataset.py:103-116 - the only one main.py uses, main.py:171-197 - and new_length 1): the dataset lives in HBM
 * as one packed fp32 store [total_frames, feature_dim] (ta3n_amd/feature_store.py writes it from the reference's
 * one-.t7-file-per-frame layout); video i owns rows [first_row[i], first_row[i] + num_frames[i]).  For each of the
 * n_videos ids the segment indices are computed in float64 exactly like the reference's Python and the selected
 * rows are copied to out [n_videos * num_segments, feature_dim] - the `x` operand of ta3n_forward.  labels_out
 * [n_videos] and segment_ids_out [n_videos * num_segments] (1-based frame ids) are optional.  num_frames must be
 * >= 1 for every id (the reference fails on empty clips).  Enqueue only. */
int ta3n_gather_segments(const float *store, const int64_t *first_row, const int32_t *num_frames,
                         const int32_t *labels, const int32_t *video_ids, int n_videos, int num_segments,
                         int feature_dim, float *out, int32_t *labels_out, int32_t *segment_ids_out,
                         void *stream);

/* ---- plan ---------------------------------------------------------------------- */

/* Builds the launch plan (tile lists, workspace layout) for one configuration;
 * replaces VideoModel.__init__/_prepare_DA's shape logic (models.py:119-325).
 * Needs no GPU: device upload is deferred to the first launch. */
int ta3n_plan_create(const ta3n_config *cfg, ta3n_plan **out);
void ta3n_plan_destroy(ta3n_plan *plan);

/* Parameter table.  name is the reference state_dict key (models.py:141-294,
 * TRNmodule.py:44-54), e.g. "TRN.fc_fusion_scales.0.1.weight".  live != 0 for
 * parameters that receive a gradient in this configuration; they occupy
 * [0, ta3n_live_param_floats) of the flat buffer. */
int ta3n_num_params(const ta3n_plan *plan);
int ta3n_param_info(const ta3n_plan *plan, int index, const char **name, int64_t *offset,
                    int32_t *rows, int32_t *cols, int32_t *live);
int64_t ta3n_param_floats(const ta3n_plan *plan);
int64_t ta3n_live_param_floats(const ta3n_plan *plan);

/* Workspace: size and named regions ("F1","Y","Pr","Pv","Pf","attn","V","losses",
 * "labels","hyper","gY","gPr","gPv","gPf","g_attn", ...).  Returns -1 if unknown. */
int64_t ta3n_workspace_floats(const ta3n_plan *plan);
int64_t ta3n_ws_offset(const ta3n_plan *plan, const char *region);
int64_t ta3n_ws_size(const ta3n_plan *plan, const char *region);

/* JSON description of buffers / phases / tasks (for tests and debugging). */
int64_t ta3n_plan_describe(const ta3n_plan *plan, char *buf, int64_t cap);

/* ---- per-step entry points (enqueue only) ------------------------------------- */

/* Writes the per-step scalars into ws["hyper"]: a one-thread kernel that receives *h BY VALUE as its launch argument, in
 * stream order.  The values are captured when the call returns, so it is safe to call it any number of steps ahead of the
 * device (no staging buffer to fence).  Call every step before ta3n_forward / ta3n_train_step. */
int ta3n_set_hyper(ta3n_plan *plan, float *ws, const ta3n_hyper *h, void *stream);

/* One-time initialisation of constant workspace regions (ones vector). */
int ta3n_init_workspace(ta3n_plan *plan, float *ws, void *stream);

/* VideoModel.forward for source+target in one pass (models.py:545-722): shared
 * frame FC + dropout, frame/relation/video domain discriminators, multi-scale
 * TRN, transferable attention, video classifier.  Outputs land in ws regions
 * F1, Pf, Pr, attn, V, Y, Pv (see ta3n_ws_offset). */
int ta3n_forward(ta3n_plan *plan, const float *x, const float *params, float *ws, void *stream);

/* Loss assembly of main.train (main.py:439-451, 508-538, 559-562; loss.py:15-25):
 * reads ws["labels"] (int32 class labels of the source rows), writes
 * ws["losses"] = {total, cls, adv_rel, adv_vid, adv_frm, entropy} and the logit
 * gradients gY, gPr, gPv, gPf.  Must follow a ta3n_forward on the same ws (which
 * clears the loss scalars). */
int ta3n_loss(ta3n_plan *plan, float *ws, void *stream);

/* loss.backward() (main.py:576) from the logit gradients in ws (gY, gPr, gPv,
 * gPf and optionally g_attn) to every live parameter gradient; GradReverse
 * (models.py:20-29) is folded in as the -beta scale of the discriminator
 * input-gradient GEMMs.  grads is overwritten (not accumulated). */
int ta3n_backward(ta3n_plan *plan, const float *x, const float *params, float *grads, float *ws,
                  void *stream);

/* ta3n_forward + ta3n_loss + ta3n_backward of one step in 7 launches instead of 15: the
 * three GEMM levels of the forward pass, ONE kernel for everything between the hidden
 * activations (Hr, Hf) and their gradients (both discriminator heads, attention pooling,
 * classifier, all losses of main.py:439-562 and their backward), then three GEMM levels
 * of the backward pass.  Same workspace regions and results (up to fp32 summation order)
 * as the three separate calls; an upstream gradient in ws["g_attn"] is NOT consumed
 * (main.train never produces one).  ta3n_has_fused_step() tells whether the plan has it
 * (num_bottleneck == 256, num_class <= 64, fc_dim <= 2048); otherwise ta3n_train_step returns TA3N_ERR_INVALID. */
int ta3n_has_fused_step(const ta3n_plan *plan);
int ta3n_train_step(ta3n_plan *plan, const float *x, const float *params, float *grads, float *ws,
                    void *stream);

/* Validation bookkeeping of main.validate / test_models.py (main.py:707-735, accuracy() main.py:809-822,
 * confusion matrix test_models.py:198) on the device: after a ta3n_forward with hyper.train = 0 and beta = 0,
 * adds the cross-entropy sum, the top-1 and top-5 hit counts and the video count of the first n_videos source
 * rows (labels in ws["labels"]) to ws["metrics"][0..3] and their (label, argmax) pairs to the int32 [C][C]
 * matrix ws["confusion"]; reset != 0 clears both first.  No host synchronisation per batch. */
int ta3n_eval_metrics(ta3n_plan *plan, float *ws, int n_videos, int reset, void *stream);

/* clip_grad_norm_ + Nesterov SGD with weight decay (main.py:578-583) on the
 * flat live prefix; ws["grad_norm"] receives the pre-clip global norm. */
int ta3n_sgd_step(ta3n_plan *plan, float *params, float *grads, float *momentum, float *ws,
                  void *stream);

/* Launches [first_launch, first_launch + n_launches) of the ta3n_train_step sequence (ta3n_num_phases(plan, 4)
 * launches in all).  The last launch produces only the gradient of the shared frame FC - the first parameter
 * of the flat layout - so a data-parallel host can start the all-reduce of everything else while it runs:
 *   ta3n_train_step_range(.., 0, n - 1, ..); all-reduce grads[n1 .. live) asynchronously;
 *   ta3n_train_step_range(.., n - 1, 1, ..); all-reduce grads[0 .. n1); wait for both; ta3n_sgd_step. */
int ta3n_train_step_range(ta3n_plan *plan, const float *x, const float *params, float *grads, float *ws,
                          int first_launch, int n_launches, void *stream);

/* ta3n_gather_segments straight into rows [first_video, first_video + n_videos) x num_segments of a plan's input
 * buffer x - and, with TA3N_FLAG_BF16_STORE, into the input's bf16 twin in ws in the same pass, so no
 * ta3n_refresh_bf16 is needed for the features. */
int ta3n_gather_segments_into(ta3n_plan *plan, const float *store, const int64_t *first_row, const int32_t *num_frames,
                              const int32_t *labels, const int32_t *video_ids, int n_videos, int first_video, float *x,
                              float *ws, int32_t *labels_out, void *stream);

/* The same from a bf16 packed store (raw bf16 [total_frames, feature_dim], feature_dim % 8 == 0): the rows are copied
 * into the input's bf16 twin as they are; x may be NULL when the plan reads twins (TA3N_FLAG_BF16_STORE) - then nothing
 * fp32 is written: 2 bytes in and 2 bytes out per element instead of 4 + 4 + 2.  With x the rows are also widened. */
int ta3n_gather_segments_bf16_into(ta3n_plan *plan, const void *store16, const int64_t *first_row, const int32_t *num_frames,
                                   const int32_t *labels, const int32_t *video_ids, int n_videos, int first_video, float *x,
                                   float *ws, int32_t *labels_out, void *stream);

/* The whole-prefix update of step n with its scalars passed by value, which also leaves `next` - the per-step
 * scalars of step n+1 - in the workspace: a loop that postpones each update to the start of the next step saves
 * the separate ta3n_set_hyper upload (one host-to-device copy per step).  Arithmetic of ta3n_sgd_step[_fused]. */
int ta3n_sgd_step_next(ta3n_plan *plan, float *params, float *grads, float *momentum, float *ws, int fused_norm,
                       float lr, float momentum_coef, float weight_decay, float clip, const ta3n_hyper *next,
                       void *stream);

/* ta3n_sgd_step_next followed by ta3n_train_step, overlapped: first the update of the shared frame FC (all that the
 * step's first launch reads) together with the new scalars, then the first launch of the new step with the update
 * of every other parameter riding along as side workgroups (EPI_SGD tasks of the tile list), then the rest of the
 * step.  Same arithmetic, bit-identical results (measured: 2 us per step less than the two calls).
 * ta3n_has_pipelined_step: 1 when the plan has this sequence. */
int ta3n_has_pipelined_step(const ta3n_plan *plan);
int ta3n_train_step_after_update(ta3n_plan *plan, const float *x, float *params, float *grads, float *momentum, float *ws,
                                 int fused_norm, float lr, float momentum_coef, float weight_decay, float clip,
                                 const ta3n_hyper *next, void *stream);

/* Several pipelined steps from ONE call: for k in [0, n_steps): { optional batch assembly; ta3n_train_step_after_update with the
 * update of the step before (its learning rate: lr_pending for k = 0, hypers[k-1].lr afterwards) and next = hypers[k] }.  The
 * per-step scalars of main.train (beta, lr, dropout seeds, valid rows: main.py:350-352, 620-621, 800-802) are a function of the
 * step index that the host evaluates ahead of time; the kernels receive them by value, so the call returns as soon as the
 * launches are queued and the host is off the step's critical path (one ctypes call and 9 hipLaunchKernel per step measured
 * ~100 us of host time on a slow core against ~100 us of GPU time).  Requires a pending update (a previous
 * ta3n_train_step / ta3n_train_steps on the same buffers); the update of step n_steps - 1 stays pending with lr = hypers[n_steps-1].lr
 * (apply it with ta3n_sgd_range or the next call).  Bit-identical to n_steps single calls.  The hypers' inv_n_* must be the
 * GLOBAL counts when a communicator is given.
 * source / target (either may be NULL = the rows already in x): TSNDataSet.__getitem__ + DataLoader collation of step k on the
 * device (dataset.py:118-144), i.e. ta3n_gather_segments[_bf16]_into with video_ids[k * ids_per_step ..] before step k - carried out by extra
 * workgroups of the launch that opens step k (the update of the step before: the two jobs touch disjoint memory and both must be done before
 * the step's first GEMM launch), not by launches of their own: a fresh batch per step costs 1-3 us at the headline shape instead of 8. */
typedef struct {
    const void *store;          /* packed rows [total_frames, feature_dim]: fp32, or bf16 when bf16 != 0 */
    int32_t bf16;
    int32_t ids_per_step;       /* videos gathered per step: <= batch_source (source feed) / batch_target (target feed) */
    const int64_t *first_row;   /* device [n_videos] */
    const int32_t *num_frames;  /* device [n_videos] */
    const int32_t *labels;      /* device [n_videos]; required for the source feed (written to ws["labels"]) */
    const int32_t *video_ids;   /* device [n_steps][ids_per_step] */
} ta3n_feed;
typedef struct ta3n_comm ta3n_comm;
int ta3n_train_steps(ta3n_plan *plan, const float *x, float *params, float *grads, float *momentum, float *ws, int fused_norm,
                     float lr_pending, float momentum_coef, float weight_decay, float clip, const ta3n_hyper *hypers,
                     int n_steps, const ta3n_feed *source, const ta3n_feed *target,
                     ta3n_comm *comm /* NULL, or: ta3n_all_reduce_sum of the live gradients after every step (fused_norm must be 0) */,
                     void *scratch_bf16 /* as in ta3n_all_reduce_sum */, void *stream);

/* ta3n_train_steps for SEVERAL independent models from one call - BASELINE configs[4]: the RGB and the Flow model of a two-stream
 * run (two VideoModel instances in the reference's terms, test_models.py sums their class logits), each with its own plan,
 * buffers and HIP stream.  For k in [0, n_steps): for every job: step k of that job on the job's stream - so the queues of the
 * jobs fill at the same rate and one model's launches fill the compute units the other's launch / prologue latencies leave idle,
 * with no host call per step and stream in between.  Each job is exactly one ta3n_train_steps argument list (same requirements:
 * a pending update per job; bit-identical to calling ta3n_train_steps once per job).  Jobs must not share buffers; ordering
 * between the jobs' streams and the caller's is the caller's business (events / hipStreamWaitEvent). */
typedef struct {
    ta3n_plan *plan;
    const float *x;
    float *params, *grads, *momentum, *ws;
    int32_t fused_norm;
    float lr_pending, momentum_coef, weight_decay, clip;
    const ta3n_hyper *hypers;       /* [n_steps] */
    const ta3n_feed *source, *target;
    ta3n_comm *comm;
    void *scratch_bf16;
    void *stream;
} ta3n_steps_job;
int ta3n_train_steps_multi(const ta3n_steps_job *jobs, int n_jobs, int n_steps);

/* The same steps with the optimiser INSIDE the gradient launches: every gradient tile of ta3n_train_step applies the Nesterov /
 * weight-decay update (main.py:83, 583) to its own block of parameters in its epilogue - the gradient is in registers, the old
 * parameter and momentum entries are read once, the new ones written once - instead of a separate pass over 5 x 4 B per parameter
 * after the step.  New parameters go to the OTHER of two buffers (params / params_alt alternate; launches of the same step that
 * still read the old values are not disturbed), the bf16 twins likewise ("p16" / "p16b").  clip_grad_norm_ (main.py:578-581)
 * needs the norm of the whole gradient, known only after the last launch: the tiles update with the clip coefficient taken as
 * 1 and one short launch per step (which also delivers the next step's scalars) checks the norm and, if it exceeded `clip`,
 * corrects parameters and momentum exactly (the update is linear in the gradient).  Bit-identical to ta3n_train_steps whenever
 * no step clips; within fp32 rounding of the correction otherwise.  Single rank (a gradient all-reduce would have to sit
 * between the tiles and the update).  Returns with the result in `params` (copied back after an odd number of steps) and nothing
 * pending; requires that no update is pending on entry.  ta3n_has_fused_update: 1 when the plan supports it (trn-m fused step). */
int ta3n_has_fused_update(const ta3n_plan *plan);
int ta3n_train_steps_fused_update(ta3n_plan *plan, const float *x, float *params, float *params_alt, float *grads, float *momentum,
                                  float *ws, float momentum_coef, float weight_decay, float clip, const ta3n_hyper *hypers,
                                  int n_steps, const ta3n_feed *source, const ta3n_feed *target, void *stream);

/* TA3N_FLAG_BF16_STORE: (re)build the bf16 twins of x (B*T*feature_dim floats, may be NULL) and of params (may be
 * NULL) inside ws.  No-op without the flag. */
int ta3n_refresh_bf16(ta3n_plan *plan, const float *x, const float *params, float *ws, void *stream);

/* Overlapping the optimiser with the next step.  The first launch of ta3n_train_step reads only x and the shared frame
 * FC (the first parameter of the flat layout), while the update is a pure HBM stream; so a host may update
 * [0, n1) (n1 = offset of the second parameter) on `stream`, put the rest of the update on a second stream, and start the
 * next step at once: ta3n_train_step_join enqueues its first launch, then makes `stream` wait for `join_event`
 * (a hipEvent_t recorded on the second stream after the rest of the update; NULL = ta3n_train_step) before the other six.
 * ta3n_sgd_range is the update over floats [begin, end) of the live prefix with the scalars passed by value (so a newer
 * ta3n_set_hyper cannot change them under it); fused_norm as in ta3n_sgd_step_fused, otherwise the range starting at 0
 * first runs the gradient-norm pass.  ws["grad_norm"] is written by the range that starts at 0. */
int ta3n_sgd_range(ta3n_plan *plan, float *params, float *grads, float *momentum, float *ws, int64_t begin,
                   int64_t end, int fused_norm, float lr, float momentum_coef, float weight_decay, float clip,
                   void *stream);
int ta3n_train_step_join(ta3n_plan *plan, const float *x, const float *params, float *grads, float *ws,
                         void *stream, void *join_event);

/* The same update directly after ta3n_train_step on the same ws, with the gradient buffer untouched in
 * between (single rank: no all-reduce): the global norm is taken from the per-tile sums of squares the
 * fused step's gradient tiles left in ws["sumsq"], which saves the pass over the gradient buffer. */
int ta3n_sgd_step_fused(ta3n_plan *plan, float *params, float *grads, float *momentum, float *ws,
                        void *stream);

/* ---- discrepancy losses (dis_DA DAN / JAN): loss.py:46-120, called from main.py:452-505 ----------------------------------------
 * ta3n_gaussian_kernel = loss.py:46-59 `guassian_kernel` on the stacked rows total = [source; target] ([n, d] fp32 device
 * memory): k_out [n, n] = sum_i exp(-||t_p - t_q||^2 / bw_i) with the reference's data-dependent bandwidth (fix_sigma <= 0) or the
 * fixed one, bw_i = bw / mul^(num / 2) * mul^i; kp_out [n, n] (may be NULL) = d k / d ||.||^2 with the bandwidth held constant
 * (loss.py:55 reads `.data`).  Distances in the explicit-difference form, fp32.  scratch: ta3n_gaussian_kernel_scratch_floats(n).
 * ta3n_mmd_rowdiff: out[p][:] = scale * sum_q c[p][q] (t[p][:] - t[q][:]) - with c = 2 kp o (gK + gK^T) the gradient of any loss at
 * the features, given its gradient gK at the kernel matrix (mmd_rbf's four-quadrant mean, JAN's product of the layers' kernels).
 * Both enqueue only; cross-workgroup sums are added in a fixed order. */
int64_t ta3n_gaussian_kernel_scratch_floats(int n);
int ta3n_gaussian_kernel(const float *total, int n, int d, float kernel_mul, int kernel_num, float fix_sigma, float *k_out, float *kp_out,
                         float *scratch, void *stream);
int ta3n_mmd_rowdiff(const float *c, const float *total, int n, int d, float scale, float *out, void *stream);

/* The whole discrepancy term of ONE rank's step, enqueued without a framework in between (main.py:452-505 between the loss and the backward):
 * kind 1 = DAN - mmd_rbf (loss.py:61-85, ver 2) of every selected feature (place_logits / place_feature: main.py's place_dis[0] / [1]) on the
 * first min(valid_source, valid_target) videos of each domain, in chunks of <= 256 videos whose losses are averaged (main.py:459-476);
 * kind 2 = JAN (loss.py:87-120, ver 2) - ONE joint kernel of logits and video feature over those videos (main.py:478-505); kernel_muls
 * [2, 2], kernel_nums [2, 5], data-dependent bandwidths.  Reads the logits [B, num_class] at ws + o_logits and the video features
 * [B, feat_dim] at ws + o_feature (source rows first); ADDS alpha * d loss / d logits to ws + o_grad_logits, WRITES alpha * d loss / d feature to
 * ws + o_grad_feature (zero rows for videos that take no part) and the loss itself (not scaled by alpha: main.py's loss_d) to *loss_out
 * (device).  Same kernels and the same gradient algebra as ta3n_gaussian_kernel / ta3n_mmd_rowdiff under autograd: gK = +-1 / half^2 per
 * quadrant (x the other layer's kernel for JAN), c = 2 Kp o (gK + gK^T).  More than one rank gathers the valid rows first
 * (ta3n_amd/parallel.py: discrepancy_over_ranks) - this entry is the single-rank path. */
/* ens_DA MCD (models.py:276-279, 716-720; main.py:447-448, 548-562; loss.py:29-30): what the reference's train loop does between the launches
 * of a step with a second classifier, without a framework in between.  A step: ta3n_forward + ta3n_loss on `ws`; ta3n_mcd_source_loss;
 * ta3n_set_hyper(reverse = 1, mu, fresh dropout seeds) + ta3n_forward on a SECOND workspace `ws2`; ta3n_mcd_second_loss; ta3n_backward on `ws`;
 * ta3n_backward on `ws2` into a second gradient buffer; the two gradients are summed.
 * ta3n_mcd_source_loss: + CrossEntropy(out_source_2, label) over the valid source rows - its logit gradient to ws["gY2"]; with
 *   TA3N_FLAG_ATTN_ENTROPY the target rows of ws["gY"] are cleared (the reference rebinds out_target to the second pass's logits before it
 *   assembles that loss, main.py:549 vs :559-562, so their entropy term belongs to the second pass).  out[0] = the loss (main.py's second loss_c term).
 * ta3n_mcd_second_loss: clears the gradient entries of ws2 (it has no ta3n_loss of its own), then over the valid target rows:
 *   loss_s = -mean |softmax(Y) - softmax(Y2)| of the second pass over (global_target x num_class) elements -> out[1], its gradients to
 *   ws2["gY"] / ws2["gY2"]; with TA3N_FLAG_ATTN_ENTROPY the target half of the attentive entropy on the second pass's logits, weighted by the
 *   FIRST pass's video-domain logits: + its gradient in ws2["gY"], ws["gPv"] moved by d(new - old) / d Pv, out[2] = what the move adds to the total
 *   loss, out[3] = to main.py's loss_e.  scratch: 3 * (batch_source + batch_target) floats; out: 4 floats (device).  Both read the step's scalars
 *   from ws["hyper"] (valid rows, inv_n_cls, gamma, inv_n_ent) and enqueue only; sums in a fixed order. */
int ta3n_mcd_source_loss(ta3n_plan *plan, float *ws, float *scratch, float *out, void *stream);
int ta3n_mcd_second_loss(ta3n_plan *plan, float *ws, float *ws2, int global_target, float *scratch, float *out, void *stream);

int64_t ta3n_discrepancy_scratch_floats(int batch_source, int batch_target, int num_class, int feat_dim);
int ta3n_discrepancy(float *ws, int64_t o_logits, int num_class, int64_t o_feature, int feat_dim, int64_t o_grad_logits, int64_t o_grad_feature,
                     int batch_source, int batch_target, int valid_source, int valid_target, int kind, int place_logits, int place_feature,
                     float alpha, float *scratch, int64_t scratch_floats, float *loss_out, void *stream);

/* ---- data parallelism: RCCL from the C ABI (replaces nn.DataParallel's per-step broadcast / gather / reduce, main.py:79) ----
 * One process per GPU.  Rank 0 makes a 128-byte id (ta3n_comm_unique_id), the launcher hands it to every rank (any
 * channel: torch.distributed store, MPI, a file), every rank calls ta3n_comm_create on its device - a collective over
 * `world` ranks (ncclCommInitRank).  RCCL is dlopen'ed at the first of these calls; a process that never calls them
 * never loads it. */
int ta3n_comm_unique_id(char *id128);
int ta3n_comm_create(const char *id128, int rank, int world, ta3n_comm **out);
void ta3n_comm_destroy(ta3n_comm *comm);
int ta3n_comm_world(const ta3n_comm *comm);

/* In-place SUM all-reduce of buf[0 .. count) over the communicator, enqueued on `stream` (ncclAllReduce).
 * scratch_bf16 == NULL: fp32 transport (exact sum of the ranks' fp32 gradients up to the reduction order).
 * scratch_bf16 != NULL (count * 2 bytes, count % 4 == 0): bf16 transport - every rank's values are rounded to bf16
 * (nearest even), summed in bf16 and widened back: half the bytes over xGMI, bf16 precision of the summed gradient. */
int ta3n_all_reduce_sum(ta3n_comm *comm, float *buf, int64_t count, void *scratch_bf16, void *stream);

/* The same exchange WITHOUT RCCL: a two-shot all-reduce over peer-mapped buffers (one process per GPU; every rank reduces one
 * 1/world chunk reading all its peers over xGMI at once, then every rank reads the reduced chunks back; csrc/ta3n_peer.hip).
 * Two crossings of 7/8 of the buffer on all 7 links in parallel instead of a ring's 14 sequential hops - what the 13.9 MB
 * exchange of this step needs to stay under the step time.  ta3n_peer_create allocates the staging buffers (fine-grained
 * device memory, for up to max_count elements, transport fp32 or bf16) on the current device; ta3n_peer_handle fills 128 bytes
 * (two HIP IPC handles) that the launcher gathers from every rank, in rank order, and hands to ta3n_peer_connect as world x 128
 * bytes.  ta3n_peer_all_reduce_sum: in-place SUM over the ranks, enqueued on `stream` (three kernels); every rank receives
 * bit-identical results; every cross-rank wait is bounded (120 s; TA3N_PEER_TIMEOUT_S in the environment, 0 = unbounded): a wait that gives
 * up sets a sticky error word, every exchange from then on delivers NaN instead of partial sums, and ta3n_peer_status (synchronises)
 * reports it - a host checks it wherever it synchronises anyway (TrainEngine.check_exchange).
 * ta3n_comm_attach_peer routes ta3n_all_reduce_sum / ta3n_train_step_ddp / ta3n_train_steps of a communicator through it. */
typedef struct ta3n_peer ta3n_peer;
int ta3n_peer_create(int rank, int world, int64_t max_count, int bf16_transport, ta3n_peer **out);
int ta3n_peer_handle(ta3n_peer *peer, char *handle128);
int ta3n_peer_connect(ta3n_peer *peer, const char *all_handles);
int ta3n_peer_all_reduce_sum(ta3n_peer *peer, float *buf, int64_t count, void *stream);
int ta3n_peer_status(ta3n_peer *peer, void *stream);
void ta3n_peer_destroy(ta3n_peer *peer);
int ta3n_comm_attach_peer(ta3n_comm *comm, ta3n_peer *peer);

/* ---- sharded optimiser step (the data-parallel exchange as reduce-scatter + all-gather instead of one all-reduce) ----------------
 * What nn.DataParallel's backward + optimizer.step() (main.py:79, 576-583) amount to when every rank updates only ITS share of the
 * parameters: the flat gradient prefix is reduce-scattered (rank r receives the job-wide SUM of its shards), every rank takes the
 * sum of squares of its shards, one float per rank is all-gathered - clip_grad_norm_'s global norm, identical on every rank -, the
 * clip + Nesterov / weight-decay update runs on the own shards only (1 / world of the optimiser's 20 bytes per parameter) and the
 * updated parameters are all-gathered.  Same bytes over xGMI as the all-reduce (which is these two collectives back to back); the
 * optimiser pass shrinks by the number of ranks and the all-gather of everything the step's first launch does not read can run
 * beside that launch.  Two regions, each dealt to the ranks in equal 4-float-aligned chunks: A = [0, a_end) holds the parameters
 * the first launch reads (the shared frame FC, whose gradient is the LAST launch's output), B = [a_end, b_end) everything else.
 * ta3n_shard_ranges: own4 = this rank's [a_lo, a_hi, b_lo, b_hi) clipped to the live prefix; layout4 = a_chunk, a_end, b_chunk, b_end.
 * ta3n_shard_sumsq: the own shards' sum of squares into slot `rank` of ws["norm_part"], every other slot zeroed (gather one float per
 * rank into slots [0, world) next).  ta3n_sgd_shard: the update on the own shards with the norm = sqrt(sum of ws["norm_part"]),
 * leaving `next` (may be NULL) as the next step's scalars.  These two need no communicator (a host may do the exchanges itself).
 * ta3n_shard_reduce_scatter / ta3n_sharded_update: the collectives (RCCL) + the pieces above on one stream; the bf16 twins of the
 * gathered parameters are rebuilt locally.  scratch_bf16 (2 bytes per element up to b_end): gradients travel as bf16. */
int ta3n_shard_ranges(const ta3n_plan *plan, int rank, int world, int64_t *own4, int64_t *layout4);
int ta3n_shard_sumsq(ta3n_plan *plan, const float *grads, float *ws, int rank, int world, void *stream);
int ta3n_sgd_shard(ta3n_plan *plan, float *params, float *grads, float *momentum, float *ws, int rank, int world, float lr,
                   float momentum_coef, float weight_decay, float clip, const ta3n_hyper *next, void *stream);
int ta3n_shard_reduce_scatter(ta3n_plan *plan, ta3n_comm *comm, float *grads, void *scratch_bf16, void *stream);
int ta3n_sharded_update(ta3n_plan *plan, ta3n_comm *comm, float *params, float *grads, float *momentum, float *ws, float lr,
                        float momentum_coef, float weight_decay, float clip, const ta3n_hyper *next, void *stream);
/* ta3n_train_steps with the sharded update: for every step { batch assembly; update of the step before (its gradients must already
 * be reduce-scattered: a previous call of this function or ta3n_shard_reduce_scatter) carrying hypers[k]; all-gather of region A on
 * `stream`, of region B on `comm_stream` under the step's first launch; the step's launches; reduce-scatter of region B on
 * `comm_stream` beside the last launch, of region A behind it }.  comm_stream == NULL or == stream: everything in order on one
 * stream.  The update of step n_steps - 1 stays pending, reduce-scattered (finish with ta3n_sharded_update, next = NULL). */
int ta3n_train_steps_sharded(ta3n_plan *plan, ta3n_comm *comm, const float *x, float *params, float *grads, float *momentum, float *ws,
                             float lr_pending, float momentum_coef, float weight_decay, float clip, const ta3n_hyper *hypers, int n_steps,
                             const ta3n_feed *source, const ta3n_feed *target, void *scratch_bf16, void *stream, void *comm_stream);

/* ta3n_train_step followed by the all-reduce of the live gradient prefix - the data-parallel step up to the optimiser
 * (continue with ta3n_sgd_step, whose norm pass reads the reduced gradients).  Losses must be normalised by GLOBAL row
 * counts (ta3n_hyper.inv_n_*), so the summed gradients are the global-batch gradients.
 * comm_stream == NULL or == stream: everything on one stream, ONE collective after the last launch (no events).
 * comm_stream != stream: the gradients of everything but the shared frame FC (complete before the last launch) are
 * reduced on comm_stream while the last launch runs, the shared frame FC's afterwards; `stream` continues after
 * both (three event edges, all inside this call; capturable into one hipGraph with the kernels). */
int ta3n_train_step_ddp(ta3n_plan *plan, ta3n_comm *comm, const float *x, const float *params, float *grads, float *ws,
                        void *scratch_bf16, void *stream, void *comm_stream);

/* Measurement aid: per-launch durations (ms) of every phase of one train step,
 * taken with HIP events recorded on `stream`; GEMM/pool/loss phases are repeated
 * `reps` times back to back.  kind_out[i]: 0 GEMM, 1 pool fwd, 2 loss, 3 pool bwd,
 * 4 grad norm, 5 SGD, 6 fused heads; group_out[i]: 0 forward, 1 loss, 2 backward,
 * 3 optimiser, 4 fused step (ta3n_train_step).  Synchronises the stream.  Returns the
 * number of phases. */
int ta3n_time_phases(ta3n_plan *plan, const float *x, float *params, float *grads, float *momentum,
                     float *ws, void *stream, int reps, float *ms_out, int32_t *kind_out,
                     int32_t *group_out, int cap);

/* Measurement aid for the pipelined step: ms_out2[0] = the optimiser launch that opens the step (update of the shared frame
 * FC + the new scalars), ms_out2[1] = the first GEMM launch with the rest of the update riding as side workgroups - the
 * two launches ta3n_train_step_after_update runs instead of the plain first launch that ta3n_time_phases times.  APPLIES
 * the update `reps` times (use it at the end of a run).  Synchronises the stream. */
int ta3n_time_update_launches(ta3n_plan *plan, const float *x, float *params, float *grads, float *momentum, float *ws,
                              int fused_norm, float lr, float momentum_coef, float weight_decay, float clip, void *stream,
                              int reps, float *ms_out2);

/* Number of kernel launches the last ta3n_forward/ta3n_backward enqueued, and a
 * name for the dominant GEMM kernel symbol (for rocprof matching). */
int ta3n_num_phases(const ta3n_plan *plan, int which /*0 fwd,1 loss,2 bwd,3 sgd,4 fused step*/);

/* Test/debug access to the plan's host-side descriptor arrays (Seg/Task/Phase/
 * Geom PODs of ta3n_amd/csrc/ta3n_types.h) so the wiring can be validated on a
 * machine without a GPU (tests/plan_interp.py). */
int ta3n_debug_arrays(const ta3n_plan *plan, const void **segs, int64_t *n_segs, const void **tasks,
                      int64_t *n_tasks, const void **phases, int64_t *n_phases, const void **geom,
                      const int32_t **tuples, const int32_t **tuple_first);
int ta3n_debug_struct_sizes(int32_t *seg, int32_t *task, int32_t *phase, int32_t *geom, int32_t *hyper);
/* ... and to the wait lists of the chained launches (ta3n_config.chain): {counter, target} int32 pairs. */
int ta3n_debug_waits(const ta3n_plan *plan, const void **waits, int64_t *n_waits);

/* ta3n_config.chain: 0 when every chained launch enqueued so far on `stream` ran to completion with all hand-offs served; 1
 * (with a message in ta3n_last_error) when a workgroup gave up waiting or a launch left its counters dirty.  Synchronises
 * the stream: a check for tests and for the end of a run, not for the step loop. */
int ta3n_chain_status(ta3n_plan *plan, const float *ws, void *stream);

const char *ta3n_last_error(void);
const char *ta3n_version(void);

#ifdef __cplusplus
}
#endif
#endif /* TA3N_HIP_H */
