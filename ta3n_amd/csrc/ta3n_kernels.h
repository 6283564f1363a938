// Device-side helpers and launcher declarations shared by the .hip files.
#pragma once
// -DTA3N_EXPERIMENTS=1 (the A/B build: TA3N_LIBDIR=ta3n_amd/lib_ab TA3N_EXTRA_FLAGS=-DTA3N_EXPERIMENTS=1 python -m ta3n_amd.build; tests:
// `pytest -m gpu_ab` with the same TA3N_LIBDIR) keeps the step-level variants that were built, measured and rejected - chained launches
// (ta3n_config.chain), split-K tiles (split_k), time-based tile order (cost_model), the optimiser inside the gradient tiles
// (ta3n_train_steps_fused_update), the four-wave 192x128 / 256x128 and four-half-stage kernels.  The default library carries only what a
// plan can select by default.  ta3n_plan_create still BUILDS such launch lists (host code, checked on the CPU by tests/plan_interp.py); launching one
// on the default library fails with TA3N_ERR_INVALID and a message naming the flag (launch_gemm returns -5, every caller maps it).  History: docs/history/.
#ifndef TA3N_EXPERIMENTS
#define TA3N_EXPERIMENTS 0
#endif
#include <hip/hip_runtime.h>

#include "ta3n_types.h"

namespace ta3n {

struct Ptrs {
    const float *x;   // BASE_X  input features [B*T, D]
    const float *p;   // BASE_P  flat parameters
    float *g;         // BASE_G  flat gradients
    float *ws;        // BASE_WS workspace
    const float *p16; // BASE_P16 bf16 twins of the parameters this launch reads (inside ws; nullptr: no twins)
};

// Side job of a GEMM launch (EPI_SGD tasks): the optimiser update of a parameter range, with its scalars by value.
// ta3n_train_step_after_update puts the update of everything but the shared frame FC into the first launch of the
// next step, whose own tiles only read the shared frame FC.
struct SgdSide {
    float *params, *momentum;
    float lr, mu, wd, clip;
    int32_t norm_off, norm_n;    // ws offsets of the squared-norm partials (fused per-tile slots or the grad_norm kernel's)
    int32_t p16_off;             // ws offset of the parameter twins, -1: none
    // Fused update (ta3n_train_steps_fused_update): every gradient tile applies the Nesterov step to its own block of parameters in
    // its epilogue - reads the parameters the step computes with (Ptrs.p) and the momentum, writes the NEW parameters (and their
    // twins) into the other buffer, so launches of the same step that still read the old values are not disturbed.  The clip
    // coefficient is taken as 1; sgd_fixup_kernel corrects the rare step whose gradient norm exceeds the clip value.
    float *p_new;                // nullptr: no fused update
    float *p16_new;              // twin region of p_new (floats), nullptr: no twins
};

__device__ __forceinline__ float hyper_scale(const Hyper *__restrict__ hy, int kind) {
    switch (kind) {
        case SK_NEG_BETA_REL: return -hy->beta[0];
        case SK_NEG_BETA_VID: return -hy->beta[1];
        case SK_NEG_BETA_FRM: return -hy->beta[2];
        case SK_INV_KEEP_I: return (hy->train && hy->p_drop_i > 0.f) ? (hy->p_drop_i < 1.f ? 1.f / (1.f - hy->p_drop_i) : 0.f) : 1.f;
        case SK_INV_KEEP_V: return (hy->train && hy->p_drop_v > 0.f) ? (hy->p_drop_v < 1.f ? 1.f / (1.f - hy->p_drop_v) : 0.f) : 1.f;
        case SK_REVERSE_MU: return hy->reverse ? -hy->mu : 1.f;
        default: return 1.f;
    }
}

// Stateless dropout stream: keep(seed, element) is recomputed in the backward
// pass instead of storing a mask (nn.Dropout, reference models.py:131-132,
// 574-575, 679-680; bit-matching torch's Philox/MT streams is not possible nor
// required - SURVEY.md 7 "Dropout").
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float keep_mask(uint32_t seed, uint32_t idx, float p) {
    const uint32_t h = mix32(mix32(idx + 0x9E3779B9U * (seed | 1u)) ^ seed);
    const float u = (float)(h >> 8) * (1.0f / 16777216.0f);   // [0,1)
    return u >= p ? 1.f : 0.f;
}

// Wave64 all-reduce on the DPP network (row-local butterflies, then row_bcast 15 / 31, then a readlane of lane 63):
// ~13 VALU instructions.  The __shfl_xor form compiles to six dependent ds_bpermute_b32 round trips through
// the LDS crossbar, which dominated the latency of the per-video stages of the heads kernel.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move(float identity, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, identity), __builtin_bit_cast(int, v),
                                                                  CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_allreduce_sum(float v) {
    v += dpp_move<0xB1, 0xF>(0.f, v);    // quad_perm [1,0,3,2]
    v += dpp_move<0x4E, 0xF>(0.f, v);    // quad_perm [2,3,0,1]
    v += dpp_move<0x141, 0xF>(0.f, v);   // row_half_mirror
    v += dpp_move<0x140, 0xF>(0.f, v);   // row_mirror: every lane of a 16-lane row holds the row sum
    v += dpp_move<0x142, 0xA>(0.f, v);   // row_bcast:15 into rows 1 and 3
    v += dpp_move<0x143, 0xC>(0.f, v);   // row_bcast:31 into rows 2 and 3: lanes 48-63 hold the total
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// This thread's share of  sum_{k < n} src[k]  (elements tid, tid + stride, ...), added in exactly the order of the plain loop
//     for (k = tid; k < n; k += stride) acc += src[k];
// but with EIGHT loads in flight per round trip: the plain loop compiled to load - s_waitcnt vmcnt(0) - add per iteration, i.e. n / stride
// dependent L2 round trips (five to twelve for the ~1 200 - 3 000 gradient-norm partials of a step) at the top of the optimiser launch
// that opens every step and of each of the update's side workgroups (ISA pass of round 5).  Elements past n are added as + 0.f:
// the partial sums are sums of squares (>= +0), so acc + 0 = acc bit for bit.
__device__ __forceinline__ float strided_partial_sum(const float *__restrict__ src, int n, int tid, int stride) {
    float acc = 0.f;
#pragma unroll 1
    for (int k0 = tid; k0 < n; k0 += 8 * stride) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = k0 + i * stride;
            v[i] = k < n ? src[k] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += v[i];
    }
    return acc;
}

__device__ __forceinline__ float wave_allreduce_max(float v) {
    const float ninf = -INFINITY;
    v = fmaxf(v, dpp_move<0xB1, 0xF>(ninf, v));
    v = fmaxf(v, dpp_move<0x4E, 0xF>(ninf, v));
    v = fmaxf(v, dpp_move<0x141, 0xF>(ninf, v));
    v = fmaxf(v, dpp_move<0x140, 0xF>(ninf, v));
    v = fmaxf(v, dpp_move<0x142, 0xA>(ninf, v));
    v = fmaxf(v, dpp_move<0x143, 0xC>(ninf, v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

struct Soft2 {
    float p0, p1, lp0, lp1, H;
};
// softmax / log_softmax / entropy of a 2-vector, same formulas as torch
// (x - max, exp, sum; log_softmax = x - max - log(sum)).
__device__ __forceinline__ Soft2 soft2(float z0, float z1) {
    Soft2 s;
    const float m = fmaxf(z0, z1);
    const float e0 = expf(z0 - m), e1 = expf(z1 - m);
    const float sum = e0 + e1;
    s.p0 = e0 / sum; s.p1 = e1 / sum;
    const float ls = logf(sum);
    s.lp0 = z0 - m - ls; s.lp1 = z1 - m - ls;
    s.H = -(s.p0 * s.lp0 + s.p1 * s.lp1);
    return s;
}

bool tile_config_ok(int cfg);   // WM*100 + WN*10 + WK of an instantiated gemm_tiles<WM, WN, WK>
int launch_gemm(const Phase &ph, const Task *d_tasks, const Seg *d_segs, const Ptrs &ptrs, int hyper_off,
                int zeros_off, int twin_off, hipStream_t stream, const SgdSide *side = nullptr, const Wait *d_waits = nullptr,
                int pair_delta = 0, int kinds = 0);
#if defined(__HIPCC__)
// two floats -> one dword of two bf16 (round to nearest even: v_cvt_pk_bf16_f32), low half = first argument
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
// the lo halves of the split x = hi + lo (pair twins, Geom::pair_delta): lo = bf16(x - float(hi)); the subtraction is exact in fp32
__device__ __forceinline__ unsigned pack_bf16_lo(float x0, float x1, unsigned hi) {
    return pack_bf16(x0 - __builtin_bit_cast(float, hi << 16), x1 - __builtin_bit_cast(float, hi & 0xFFFF0000u));
}
#endif

int launch_to_bf16(const float *src, float *dst_twin, int64_t n, hipStream_t stream);   // n fp32 -> n bf16 (RNE), n % 4 == 0
int launch_to_bf16_pair(const float *src, float *dst_hi, float *dst_lo, int64_t n, hipStream_t stream);   // ... and the lo plane
bool heads_supported(int NB, int C, int F);   // configurations the fused heads kernel (ta3n_heads.hip) covers
int launch_heads(const Geom &g, const Ptrs &ptrs, hipStream_t stream);
int launch_pool_cls(const Geom &g, const Ptrs &ptrs, hipStream_t stream);   // TA3N_AGG_AVGPOOL: between F1 and gZ1
int launch_pool_avg_fwd(const Geom &g, const Ptrs &ptrs, hipStream_t stream);
int launch_pool_avg_bwd(const Geom &g, const Ptrs &ptrs, hipStream_t stream);
int launch_bn_shared_fwd(const Geom &g, const Ptrs &ptrs, hipStream_t stream);
int launch_mcd_source_loss(const Geom &g, float *ws, float *part, float *out, hipStream_t stream);                            // ens_DA MCD glue (ta3n_mcd_*)
int launch_mcd_second_loss(const Geom &g, float *ws, float *ws2, float inv_count, float *part, float *out, hipStream_t stream);
int launch_bn_shared_bwd(const Geom &g, const Ptrs &ptrs, hipStream_t stream);
int launch_pool_fwd(const Geom &g, const Ptrs &ptrs, hipStream_t stream);
int launch_loss(const Geom &g, const Ptrs &ptrs, hipStream_t stream);
int launch_pool_bwd(const Geom &g, const Ptrs &ptrs, hipStream_t stream);
int launch_grad_norm(const Geom &g, const float *grads, float *ws, hipStream_t stream);
int launch_sgd(const Geom &g, float *params, const float *grads, float *momentum, float *ws, hipStream_t stream,
               bool fused_norm = false);
int launch_fill(float *dst, float value, int64_t n, hipStream_t stream);
int launch_set_hyper(float *ws_hyper, const Hyper &h, hipStream_t stream);   // scalars by kernel argument (no staging copy)
int launch_sgd_fixup(const Geom &g, float *params, const float *grads, float *momentum, float *ws, float *p16, float lr, float mu,
                     float clip, const Hyper *next, hipStream_t stream);
int launch_sgd_range(const Geom &g, float *params, const float *grads, float *momentum, float *ws, int64_t begin, int64_t end,
                     bool fused_norm, float lr, float mu, float wd, float clip, const Hyper *next, hipStream_t stream, bool write_norm = false);
// One batch half of the launch that opens a pipelined step together with its batch assembly (sgd_open_feed_kernel): device pointers of a packed
// feature store, this step's video ids, where the rows go (fp32 rows / bf16 twin rows; either may be null) - n_videos = 0: nothing to assemble
struct FeedJob {
    const void *store;
    const int64_t *first_row;
    const int32_t *num_frames, *labels, *video_ids;
    float *out, *twin;
    int32_t *labels_out;
    int32_t n_videos, bf16;
};
int launch_sgd_open_feed(const Geom &g, float *params, const float *grads, float *momentum, float *ws, int64_t begin, int64_t end,
                         bool fused_norm, float lr, float mu, float wd, float clip, const Hyper *next, const FeedJob feeds[2], hipStream_t stream);
int launch_shard_sumsq(const Geom &g, const float *grads, float *ws, int64_t a0, int64_t a1, int64_t b0, int64_t b1, int rank, hipStream_t stream);
int launch_eval_metrics(const Geom &g, float *ws, int n, int reset, hipStream_t stream);
int launch_gather_segments(const float *store, const int64_t *first_row, const int32_t *num_frames, const int32_t *labels,
                           const int32_t *video_ids, int n_videos, int T, int D, float *out, int32_t *labels_out, int32_t *seg_out,
                           float *out_twin, hipStream_t stream, int64_t pair_delta = 0);

int launch_gather_segments_bf16(const void *store16, const int64_t *first_row, const int32_t *num_frames, const int32_t *labels,
                                const int32_t *video_ids, int n_videos, int T, int D, float *out, int32_t *labels_out, float *out_twin,
                                hipStream_t stream, int64_t pair_delta = 0);


}  // namespace ta3n

// pieces of the sharded update that live beside the RCCL binding (csrc/ta3n_comm.hip)
struct ta3n_comm;
namespace ta3n {
int comm_rank(const ta3n_comm *c);
int comm_world_size(const ta3n_comm *c);
hipEvent_t comm_event(ta3n_comm *c, int which);      // 0 fork, 1 join, 2 fork2, 3 join2 (created on first use)
int comm_reduce_scatter_sum(ta3n_comm *c, float *buf, int64_t begin, int64_t chunk, void *scratch_bf16, hipStream_t s);
int comm_all_gather(ta3n_comm *c, float *buf, int64_t begin, int64_t chunk, hipStream_t s);
}  // namespace ta3n
