// Internal POD descriptors shared by the host-side plan builder and the device
// kernels.  A "phase" is one kernel launch; a GEMM phase runs a list of tile
// tasks, each task a list of K-segments.  Everything addresses memory as
// (base id, element offset) so a plan is independent of buffer addresses.
#pragma once
#include <stdint.h>

namespace ta3n {

// BASE_P16: the bf16 twins of the parameters (TA3N_FLAG_BF16_STORE), addressed in floats from the start of the twin region the
// launch is given (Ptrs.p16): the fused-update step alternates between two parameter buffers and their twin regions
enum Base : int32_t { BASE_NONE = -1, BASE_X = 0, BASE_P = 1, BASE_G = 2, BASE_WS = 3, BASE_P16 = 4, BASE_COUNT = 5 };

// segment / epilogue scale kinds: value is taken from the device Hyper struct
enum ScaleKind : int32_t {
    SK_ONE = 0,
    SK_NEG_BETA_REL = 1,   // -beta[0]   GradReverse on relation features  (models.py:474-476)
    SK_NEG_BETA_VID = 2,   // -beta[1]   (models.py:465)
    SK_NEG_BETA_FRM = 3,   // -beta[2]   (models.py:457)
    SK_INV_KEEP_I = 4,     // 1/(1-p_drop_i) when training else 1
    SK_INV_KEEP_V = 5,     // 1/(1-p_drop_v) when training else 1
    SK_REVERSE_MU = 6,     // -mu when hyper.reverse else 1   (models.py:682-684)
};

enum EpiFlags : uint32_t {
    EPI_BIAS = 1u << 0,     // v += bias[n]
    EPI_ADD = 1u << 1,      // v = alpha*v + add[m][n]
    EPI_RELU = 1u << 2,     // v = max(v, 0)
    EPI_MASK = 1u << 3,     // v *= (aux[m][n] > 0)
    EPI_DROP_I = 1u << 4,   // v *= keep_i(m*drop_ld + n)   (dropout_i stream)
    EPI_DROP_V = 1u << 5,   // v *= keep_v(...)
    EPI_SUMROWS8 = 1u << 6, // workgroup side job: ws[pad[0] + c] = sum_r ws[pad[1] + 8 r + c], r < pad[2], c < 8 (loss scalars of the fused step)
    EPI_COLSUM = 1u << 12,  // not a tile: c[n] = sum_{r < pad[1]} ws[pad[0] + r * pad[2] + n] for n in [n0, n_valid), rows added in order by
                            // one thread per column - exact fp32 reduction of per-workgroup partial sums (never through the MFMA,
                            // which would round the partials to bf16 in the bf16 configuration); honours EPI_SUMSQ
    EPI_TWIN_ONLY = 1u << 13,      // with EPI_TWIN16: do not store the fp32 tile at all (nobody reads it: every consumer reads the twin)
    EPI_TWIN_ONLY_FAN = 1u << 14,  // same for the fan-out copies
    EPI_SPLITK = 1u << 15,  // the tile is computed by TWO tasks of the launch, each over part of its K segments (ta3n_config.split_k): pad[2] - 1 =
                            // this task's half, pad[0] = ws offset of the pair's partial tiles [2][BM x BN], pad[1] = ws offset of its
                            // ticket (int32, zero between launches).  Both publish their partial and take a ticket; the second adds the
                            // other's partial and runs the epilogue, the first is done.
    EPI_SGD = 1u << 11,     // not a tile: the workgroup applies the optimiser update to params[4 pad[0] .. 4 pad[1]) (SgdSide)
    EPI_TWIN16_FAN = 1u << 10,   // same for the fan-out copies (fan_out_off)
    EPI_TWIN16 = 1u << 9,   // also store the tile rounded to bf16 at ws16 + (c_off + m * c_ld + n) * 2 bytes (c_base == BASE_WS)
    EPI_ROWSUM_A = 1u << 8, // also store the K-sums of the tile's A rows to bias_base[bias_off + m] (bias gradient of a weight-gradient tile)
    EPI_SUMSQ = 1u << 7,    // workgroup side job: ws[pad[3]] = sum of squares of the stored tile (fused grad-norm partial)
};

struct Seg {
    int32_t a_base, b_base;
    int32_t a_off, b_off;        // element offsets of the operand origin
    int32_t a_ld, b_ld;          // K-contiguous operand: row stride; k-major operand: stride between k rows
    int32_t a_kmajor, b_kmajor;  // 0: element (r,k) at off + r*ld + k ; 1: at off + k*ld + r
    int32_t klen;
    int32_t scale_kind;          // accumulator *= scale after this segment (SK_ONE = none)
    int32_t pad[2];              // [0] > 0: readable rows of the A operand (overrides the task's m_valid for loading)
};

struct Task {
    int32_t m0, n0;              // tile origin in the output
    int32_t m_valid, n_valid;    // output extent (rows / cols) for guards
    int32_t seg_begin, seg_count;
    uint32_t epi;                // EpiFlags
    int32_t alpha_kind;          // ScaleKind applied to the accumulator before `add`
    int32_t gamma_kind;          // ScaleKind applied last
    int32_t c_base, c_off, c_ld;
    int32_t bias_base, bias_off;
    int32_t aux_base, aux_off, aux_ld;   // mask operand
    int32_t add_base, add_off, add_ld;   // additive operand
    int32_t drop_ld;                     // element id for dropout = m*drop_ld + n
    int32_t fan_count;                   // extra masked outputs (all in BASE_WS, ld = fan_ld)
    int32_t fan_ld;
    int32_t fan_mask_off[3];
    int32_t fan_out_off[3];
    Seg seg0;                            // copy of segs[seg_begin]: arrives with the Task, so the first DMA is one dependent
                                         // scalar load away from kernel entry instead of two
    int32_t cost;                        // sum of klen (for ordering / balance)
    int32_t pad[4];                      // EPI_SUMROWS8: [0..2] = {dst, src, rows} (ws offsets); EPI_SUMSQ: [3] = ws offset of the slot;
                                         // EPI_COLSUM: [0..2] = {src ws offset, rows, row stride}
    // Chained launch (Phase.chain_off >= 0): several dependency levels in ONE launch.  A task whose output a later task of the
    // same launch reads increments counter `sig` once its stores are visible at agent scope; a task that reads such output first
    // waits until every (counter, target) pair of its wait list waits[wait_begin .. wait_begin + wait_count) is reached.  The plan
    // builder derives both from the tasks' read / write spans (ta3n_plan.cpp: derive_chain); tasks only ever wait on tasks with
    // a lower index in the launch.
    int32_t sig;                         // -1: nobody in this launch reads what this task writes
    int32_t wait_begin, wait_count;
    int32_t pad2;
};
struct Wait { int32_t counter, target; };

enum PhaseKind : int32_t {
    PH_GEMM = 0,
    PH_POOL_FWD = 1,
    PH_LOSS = 2,
    PH_POOL_BWD = 3,
    PH_GRAD_NORM = 4,
    PH_SGD = 5,
    PH_HEADS = 6,            // fused video/frame heads: forward + loss + backward between Hr/Hf and gHr/gHf
    PH_POOL_CLS = 7,         // TA3N_AGG_AVGPOOL, source-only fused: mean over segments, dropout, classifier, CE and the way back to gZ1
    PH_POOL_AVG_FWD = 8,     // TA3N_AGG_AVGPOOL, general: V = mean over the segments of F1, Vd = dropout_v(V)
    PH_POOL_AVG_BWD = 9,     // TA3N_AGG_AVGPOOL, general: gradient at V spread back over the segments (-> gZ1 or its additive base)
    PH_BN_FWD = 10,          // TA3N_FLAG_BN_SHARED: F1 = dropout_i(relu(BatchNorm_domain(Z0)))
    PH_BN_BWD = 11,          // ... and back: gZ0, d(bn weight), d(bn bias) from gZ1
};

// work split of the fused heads kernel (ta3n_heads.hip); the plan builder sizes its partial-sum regions from these
constexpr int HEADS_RPW = 16;   // frame rows per row group of a frame workgroup (Geom.heads_rpw rows per workgroup, a multiple of this)

struct Phase {
    int32_t kind;
    int32_t group;               // 0 fwd, 1 loss, 2 bwd, 3 sgd, 4 fused fwd+loss+bwd (ta3n_train_step)
    int32_t task_begin, task_count;
    int32_t wm, wn, wk;          // wave grid of the GEMM tile (block tile = 32*wm x 32*wn, wk-way K split)
    int32_t bf16;                // 0: fp32 MFMA; 2 / 3: operands rounded to bf16 for the MFMA (TA3N_FLAG_BF16_MFMA), LDS stages;
                                 // + 32: TA3N_FLAG_F32_SPLIT - operands split hi + lo in registers, three bf16 MFMAs per product block;
                                 // + 16: the operands ARE bf16 (TA3N_FLAG_BF16_STORE): the Segs' offsets address the
                                 // bf16 twins (in floats, base BASE_WS); ld, klen and row counts stay in elements
                                 // + 16 + 32: pair twins - the stage holds the hi AND the lo plane of 64 k (Geom::pair_delta), three
                                 // MFMAs per product block, no conversion in the loop
    int32_t rm, rn;              // 32x32 blocks per wave (0 or 1: one): block tile = 32*wm*rm x 32*wn*rn; > 1 only when bf16 >= 16
    int32_t chain_off;           // chained launch: ws offset (floats) of its int32 block {done, error, counters[chain_n]}; -1: plain launch
    int32_t chain_n;
};

// Mirror of ta3n_hyper (include/ta3n_hip.h); the device reads it from ws.
struct Hyper {
    float beta[3];
    float gamma, lr, momentum, weight_decay, clip;
    float p_drop_i, p_drop_v;
    uint32_t seed_i, seed_v;
    float inv_n_cls, inv_n_rel, inv_n_vid, inv_n_frm, inv_n_ent;
    int32_t valid_source, valid_target, train;
    int32_t reverse;             // forward(..., reverse=True): GradReverse(mu) between dropout_v and the video heads (models.py:682-684)
    float mu;
    int32_t reserved[2];
};

// Columns per workgroup of the two BatchNorm launches (TA3N_FLAG_BN_SHARED): a thread moves the BN_COLS columns of one row as one 16-byte
// access; the backward launch leaves one gradient-norm slot per workgroup, so the plan builder needs the number too.
constexpr int BN_COLS = 4;

// Device-side constant geometry handed to the pointwise kernels by value.
struct Geom {
    int32_t Bs, Bt, B, T, D, F, NB, C;
    int32_t n_tuples;            // total relation tuples
    int32_t n_rel;               // T-1
    uint32_t flags;
    // workspace offsets (elements)
    int32_t o_F1, o_Hf, o_Pf, o_Zr, o_Hr, o_Pr, o_R, o_attn, o_V, o_Vd, o_Y, o_Hv, o_Pv;
    int32_t o_gY, o_gPv, o_gPr, o_gPf, o_gattn, o_gHv, o_gHf, o_gVt, o_gPrT, o_gRa, o_gHr, o_gR, o_gZ, o_gZ1;
    int32_t o_zeros, o_ones, o_losses, o_norm_part, o_grad_norm, o_hyper, o_labels, o_tuple_first;
    int32_t n_norm_blocks;
    int32_t live_floats;
    // relation discriminator second layers inside the flat parameter buffer
    int32_t p_W2_0, p_b2_0, p_W2_stride, p_b2_stride;   // W2_j at p_W2_0 + j*p_W2_stride
    // fused heads (ta3n_heads.hip)
    int32_t p_Wcd, p_bcd, p_Wdv, p_bdv, p_Wcv, p_bcv, p_Wcdv, p_bcdv;
    int32_t o_fh_part, o_fh_bpart;       // per frame-workgroup partial sums of dWcd [n_frm_wg][2F] and dbcd [n_frm_wg][2]
    int32_t o_loss_part;                 // per heads-workgroup loss partials [n_vid_wg + n_frm_wg][8]
    int32_t n_vid_wg, n_frm_wg;
    int32_t heads_rpw;                   // frame rows per frame workgroup of the fused heads kernel (multiple of HEADS_RPW)
    int32_t o_sumsq, n_sumsq;            // fused grad-norm partials (one slot per gradient tile of the fused step)
    int32_t o_metrics, o_confusion;      // validation: {sum CE, top-1 hits, top-5 hits, videos} and the int32 [C][C] confusion matrix
    // bf16 twins (TA3N_FLAG_BF16_STORE), all inside ws, offsets in floats: element e of ws / params / x lives, rounded
    // to bf16, at byte (o_*16 * 4 + 2 e).  -1: no twins.
    int32_t o_ws16, o_p16, o_x16, ws16_span;   // ws16 mirrors ws[0 .. ws16_span)
    int32_t o_gV_ext;                    // TA3N_FLAG_FEATURE_GRADS: caller-written gradient at V, added to gVt by the pooling backward (0: none)
    int32_t o_Y2, o_gY2;                 // TA3N_FLAG_MCD: second classifier's logits / logit gradients (0: none)
    // TA3N_FLAG_BN_SHARED: linear output before / gradient behind the domain BatchNorm, batch and running statistics, parameters
    int32_t o_Z0, o_gZ0, o_bn_batch, o_bn_run;
    int32_t p_bn_w[2], p_bn_b[2];        // [source, target]
    int32_t o_p16b;                      // second parameter-twin region (fused-update step: ping-pong with the second parameter buffer); -1: none
    // TA3N_FLAG_F32_SPLIT with twins ("pair twins"): every twin region has a second plane holding lo = bf16(x - float(hi)) beside
    // hi = bf16(x); the lo plane of ANY twin element lives pair_delta floats behind its hi plane (the four lo regions are laid out
    // like the four hi regions).  0: no lo planes.
    int32_t pair_delta;
    int32_t heads_vpw;                   // videos per video workgroup of the fused heads kernel (1 or 2; 0 = 1)
};

// Register-blocking digit of a tile code (ten-thousands): 32x32 blocks per wave, rows x columns.
// 0: 1 x 1, 1: 2 x 1, 2: 1 x 2, 3: 2 x 2, 4: 3 x 2, 5: 4 x 2 (the last two: half-stage kernels of four waves, tile codes 46221 / 56221).
inline int blk_rm(int blk) { return blk == 4 ? 3 : blk == 5 ? 4 : 1 + (blk & 1); }
inline int blk_rn(int blk) { return blk >= 4 ? 2 : 1 + (blk >> 1); }
inline int blk_code(int rm, int rn) { return rm == 3 ? 4 : rm == 4 ? 5 : (rm > 1 ? 1 : 0) + (rn > 1 ? 2 : 0); }

}  // namespace ta3n
