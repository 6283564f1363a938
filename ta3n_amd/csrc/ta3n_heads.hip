// Fused "heads" kernel of the TA3N train step (gfx950, wave64): everything between
// the hidden activations (Hr, Hf) and their gradients (gHr, gHf) in ONE launch -
// forward, loss assembly and backward of
//   * the relation discriminators' 2-wide output layers, the transferable attention
//     w = 1 - H(softmax Pr), R_j = sum of the scale's tuple activations, the pooled
//     video feature V = sum_j (1+w_j) R_j and dropout_v
//     (reference models.py:472-488, 351-357, 379-388, 651, 679; TRNmodule.py:73-79),
//   * the video classifier Y and the video discriminator Hv -> Pv (models.py:686, 464-470),
//   * the frame discriminator's output layer Pf (models.py:460),
//   * classification CE, the three adversarial CEs and the attentive entropy with all
//     their logit gradients (main.py:439-451, 508-538, 559-562; loss.py:15-25),
//   * GradReverse (models.py:20-29) as the -beta factor of gVt,
//   * the un-detached attention path (dL/dw_j = <R_j, dL/dV>) back into Pr.
// It replaces seven launches of the unfused path (pool_fwd, two small GEMM levels, loss,
// two small GEMM levels, pool_bwd).
//
// Grid: n_vid_wg video workgroups (Geom::heads_vpw videos each: one wave per video for the
// per-video reductions, thread t <-> channel t for the 256-wide layers) followed by
// n_frm_wg frame workgroups (HEADS_RPW frame rows each, 4 rows per wave).
//
// The kernel is a chain of short dependent stages with one wave per SIMD, so its time
// is memory round trips, not arithmetic.  Structure follows from that:
//   * every load whose address does not depend on a computed value is issued at the top
//     of the kernel (classifier weights, first discriminator weight tile, output-layer
//     rows, biases) and lands while the first stage computes;
//   * the 256x256 video-discriminator layer is a VALU mini-GEMM over LDS-staged weight
//     tiles (forward: 256 outputs x 64 k per tile, backward: 64 outputs x 256 k), the
//     next tile travelling in registers while the current one is multiplied, shared by
//     the videos of the workgroup;
//   * a frame wave keeps its 4 rows in registers: one round trip for the forward dots,
//     none for the backward.
// Cross-workgroup sums made here (dWcd, dbcd, the logging scalars) are written as
// per-workgroup partials and added in a fixed order by the next GEMM launch, so
// results are bitwise reproducible and no atomics are used.
#include <hip/hip_runtime.h>

#include <mutex>

#include "../../include/ta3n_hip.h"
#include "ta3n_kernels.h"

using namespace ta3n;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));   // (arrays of HIP float4 are demoted to scratch by hipcc)

// (Round 6: the bisect switches of round 5's ISA pass - TA3N_HEADS_FIX - are gone.  The two changes every validated build carried are
// unconditional now: the tuple ranges of all relations by ONE vector load + v_readlane, the label by a scalar load.  The third -
// branch-free Zr loads through an address select into the block of zeros - gave wrong class logits at 12 segments together with the
// v_readlane ranges on hipcc 7.2 and was never root-caused (no documented VALU-writes-SGPR hazard in either binary:
// tools/isa_hazard_scan.py over the -save-temps ISA of all instantiations); it bought nothing measurable on its own, so the code
// is deleted rather than kept behind a switch.  Likewise the four-videos-per-workgroup instantiation, which faulted and which no
// plan ever selected.)

constexpr int NBH = 256;           // num_bottleneck of trn-m (models.py:223); thread t <-> channel t
constexpr int RPW = HEADS_RPW;
constexpr int WROW = 68;           // padded row of the forward tile [256 n][64 k] (conflict-free b128 row reads)
constexpr int TROW = 260;          // padded row of the backward tile [64 n][256 k] and of the classifier weights [C][256]

// LDS carve-up (floats) of a workgroup that handles VPW videos at once (Geom::heads_vpw: 1 where every video can have a compute unit of
// its own - the headline shape - and 2 for larger batches: a video workgroup owns its CU (256 registers of weights per lane), so
// with more videos than CUs the launch ran in ROUNDS of one video per CU, each a ~26-30 k-cycle chain of dependent single-wave stages
// (tools/heads_timing.py, round 4: 7-11 k of it in the relation stages A and G, ~13 k in B-F whatever the shape).  With VPW videos per
// workgroup the per-video stages run one video per wave (or per wave pair) side by side and the 256x256 layer multiplies VPW vectors
// from the one register copy of its weights.)
template <int VPW>
struct Lds {
    static_assert(VPW == 1 || VPW == 2, "1 or 2 videos per workgroup");
    static constexpr int W = 0;                                 // weight tile (17408 floats)
    static constexpr int VD = W + NBH * WROW;                   // [VPW][256] dropped-out video feature
    static constexpr int HV = VD + VPW * NBH;                   // [VPW][256] video-discriminator hidden
    static constexpr int GHV = HV + VPW * NBH;                  // [VPW][256] its gradient
    static constexpr int GVT = GHV + VPW * NBH;                 // [VPW][256] gradient at the pooled feature
    static constexpr int GY = GVT + VPW * NBH;                  // [VPW][64]  class-logit gradients
    static constexpr int PR = GY + VPW * 64;                    // [VPW][64][2] relation logits
    static constexpr int GPV = PR + VPW * 128;                  // [VPW][2]
    static constexpr int Y = GPV + 8;                           // [VPW][64] class logits
    static constexpr int FPART = Y + VPW * 64;                  // [16][256] partial sums of the backward mini-GEMM (one video at a time)
    static constexpr int VPART = FPART + 16 * NBH;              // [4 waves][256] per-wave partial sums of V
    static constexpr int LOSS = VPART + 4 * NBH;                // [4 waves][8] loss partials
    static constexpr int TOTAL = LOSS + 32;
    static constexpr int WPV = 4 / VPW;                         // waves per video: they split the relations of the per-video stages
};
static_assert(64 * TROW <= NBH * WROW, "backward tile / classifier staging must fit in the weight tile");
constexpr int S_LOSS_MIN = Lds<1>::LOSS;                        // (the frame workgroups' staging must stay below the loss slots of every variant)

// -DTA3N_HEADS_TIMING: the 101st video workgroup stamps s_memtime at every stage boundary into ws["g_attn"] (debug builds only)
#ifdef TA3N_HEADS_TIMING
#define STAMP(i) do { if ((int)blockIdx.x == g.n_frm_wg + 100 && threadIdx.x == 0) { reinterpret_cast<unsigned long long *>(ptrs.ws + g.o_gattn)[i] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ bool video_valid(int Bs, int valid_source, int valid_target, int b) {
    return b < Bs ? (b < valid_source) : (b - Bs < valid_target);
}

// The per-step scalars a workgroup uses, read ONCE at its top.  `hy` points into the workspace the kernel also stores to, so a read of
// hy->x inside a loop is re-issued after every store - and each such read came with s_waitcnt vmcnt(0), i.e. it also waited for the
// operands requested ahead for the NEXT relation / row (found in the ISA in round 5: one dependent round trip per relation in stage G
// and per frame row in the frame workgroups).
struct StepScalars {
    int valid_source, valid_target;
    float inv_n_cls, inv_n_rel, inv_n_vid, inv_n_frm, inv_n_ent, gamma, beta1, p_drop_v;
    uint32_t seed_v;
};
__device__ __forceinline__ StepScalars step_scalars(const Hyper *__restrict__ hy) {
    StepScalars h;
    h.valid_source = hy->valid_source; h.valid_target = hy->valid_target;
    h.inv_n_cls = hy->inv_n_cls; h.inv_n_rel = hy->inv_n_rel; h.inv_n_vid = hy->inv_n_vid; h.inv_n_frm = hy->inv_n_frm;
    h.inv_n_ent = hy->inv_n_ent; h.gamma = hy->gamma; h.beta1 = hy->beta[1]; h.p_drop_v = hy->p_drop_v; h.seed_v = hy->seed_v;
    return h;
}

// this workgroup's loss partials -> ws["loss_part"][wg][8] = {total, cls, rel, vid, frm, ent, 0, 0}
__device__ __forceinline__ void write_loss_part(float *smem, float *ws, int o_loss_part, int wg, float gamma, int S_LOSS) {
    const int tid = threadIdx.x;
    __syncthreads();
    float v = 0.f;
    if (tid < 8) v = (smem[S_LOSS + tid] + smem[S_LOSS + 8 + tid]) + (smem[S_LOSS + 16 + tid] + smem[S_LOSS + 24 + tid]);
    __syncthreads();
    if (tid < 8) smem[S_LOSS + tid] = v;
    __syncthreads();
    if (tid < 8) {
        if (tid == 0) v = smem[S_LOSS + 1] + smem[S_LOSS + 2] + smem[S_LOSS + 3] + smem[S_LOSS + 4] + gamma * smem[S_LOSS + 5];
        ws[o_loss_part + wg * 8 + tid] = v;
    }
}

// PIPE: a wave handles MORE than one relation (n_rel > waves per video; chosen at launch): the next relation's operands are requested while
// the current one is reduced, and stage G's first relation in front of stage F.  With one relation per wave (the headline shape: 5
// segments, 4 relations, 4 waves) the bookkeeping of that pipeline cost 0.5 us per launch (profiles/r05_heads_tiles_ab.txt), so the
// single-relation kernel keeps round 4's sequence: stage A's only relation requested at the top, stage G's inside stage G.
template <int VPW, bool PIPE>
__device__ __forceinline__ void video_wg(const Geom g, const Ptrs ptrs, float *smem) {
    using L = Lds<VPW>;
    constexpr int S_W = L::W, S_VD = L::VD, S_HV = L::HV, S_GHV = L::GHV, S_GVT = L::GVT, S_GY = L::GY, S_PR = L::PR, S_GPV = L::GPV, S_Y = L::Y,
                  S_FPART = L::FPART, S_VPART = L::VPART, S_LOSS = L::LOSS, WPV = L::WPV;
    float *__restrict__ ws = ptrs.ws;
    const float *__restrict__ wsr = ptrs.ws;       // regions this kernel only reads (Hr, Zr): their loads may move over stores
    const float *__restrict__ P = ptrs.p;
    const Hyper *__restrict__ hy = reinterpret_cast<const Hyper *>(ptrs.ws + g.o_hyper);
    const int *__restrict__ tf = reinterpret_cast<const int *>(ptrs.ws + g.o_tuple_first);
    const int *__restrict__ labels = reinterpret_cast<const int *>(ptrs.ws + g.o_labels);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int NR = g.n_rel, NT = g.n_tuples, C = g.C;
    const int b0 = ((int)blockIdx.x - g.n_frm_wg) * VPW;
    const int nv = min(VPW, g.B - b0);
    const int vloc = wv / WPV, sub = wv % WPV;     // per-video stages: this wave's video slot and its share of the relations
    const int b = b0 + vloc;
    const bool have = vloc < nv;
    const bool lead = sub == 0;                    // the wave that does the video's un-splittable parts (logits, losses)
    const bool attn_on = (g.flags & TA3N_FLAG_TRANS_ATTN) != 0;
    const bool train = hy->train != 0;
    const StepScalars hs = step_scalars(hy);
    const float inv_keep_v = hyper_scale(hy, SK_INV_KEEP_V);
    const bool drop_v = train && hs.p_drop_v > 0.f;
    const float *__restrict__ Wdv = P + g.p_Wdv;
    const float *__restrict__ Wcv = P + g.p_Wcv;
    float l_cls = 0.f, l_rel = 0.f, l_vid = 0.f, l_ent = 0.f;

    // ---- stage A's operands for this wave's FIRST relation, requested before anything else ----
    // Vector loads return in order: behind the 256 KB weight burst below, the first relation's 24 values per lane used to arrive last
    // and stage A - which needs nothing else - waited 4-5 k cycles for them (tools/heads_timing.py).  The tuple range comes by
    // SCALAR loads (their own counter), so the addresses are known at once; at 5 segments (4 relations, 4 waves per video) this is
    // the wave's ONLY relation.
    // (round 5) ... and inside the loop the NEXT relation's operands are requested before the current one is reduced: at 9 / 12 segments a
    // wave handles 2-3 relations (8-11 with several videos per workgroup), each of which used to be a dependent round trip of its own.
    // The tuple ranges of ALL relations come with ONE vector load at the very top (lane l holds tuple_first[l]; n_rel + 1 <= 64) and are
    // read with v_readlane afterwards: a load of tuple_first[j] inside load_rel made the compiler wait for EVERYTHING outstanding
    // (s_waitcnt vmcnt(0): the counter is in order) before it could form the next relation's addresses - a full round trip per relation
    // at the top of each loop iteration (found in the ISA in round 5).
    const int tfv = lane <= NR ? tf[lane] : 0;
    struct RelIn { float hr[4], w0[4], w1[4], zr[3][4], b0, b1; int nt; };
    auto load_rel = [&](int j, RelIn &o) {
        const float *__restrict__ W2f = P + g.p_W2_0 + (size_t)j * g.p_W2_stride;
        const float *__restrict__ hrf = wsr + g.o_Hr + ((size_t)b * NR + j) * NBH;
        const int ju = __builtin_amdgcn_readfirstlane(j);
        const int t_lo = __builtin_amdgcn_readlane(tfv, ju);
        const int t_hi = __builtin_amdgcn_readlane(tfv, ju + 1);
        o.nt = t_hi - t_lo;
#pragma unroll
        for (int q = 0; q < 4; ++q) { o.hr[q] = hrf[q * 64 + lane]; o.w0[q] = W2f[q * 64 + lane]; o.w1[q] = W2f[NBH + q * 64 + lane]; }
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) {         // a relation sums at most 3 tuples (TRNmodule.py:32 subsample_num); a tuple past the range reads as +0
            const bool on = tt < o.nt;
#pragma unroll
            for (int q = 0; q < 4; ++q) o.zr[tt][q] = on ? wsr[g.o_Zr + ((size_t)b * NT + t_lo + tt) * NBH + q * 64 + lane] : 0.f;
        }
        o.b0 = P[g.p_b2_0 + (size_t)j * g.p_b2_stride];
        o.b1 = P[g.p_b2_0 + (size_t)j * g.p_b2_stride + 1];
    };
    RelIn rel_cur;
    rel_cur.nt = 0; rel_cur.b0 = rel_cur.b1 = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) { rel_cur.hr[q] = rel_cur.w0[q] = rel_cur.w1[q] = 0.f; rel_cur.zr[0][q] = rel_cur.zr[1][q] = rel_cur.zr[2][q] = 0.f; }
    if (have && sub < NR) load_rel(sub, rel_cur);
    // ---- address-independent loads ----
    f32x4 cw[16];                                  // classifier weights [C][256]: float4 i*256+tid of the flat array
#pragma unroll
    for (int i = 0; i < 16; ++i)
        cw[i] = (i * 256 + tid < C * 64) ? *reinterpret_cast<const f32x4 *>(Wcv + (size_t)(i * 256 + tid) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    // The whole 256 x 256 video-discriminator weight matrix lives in this workgroup's registers (64 float4 per
    // thread; one wave per SIMD leaves 512 registers per lane): thread (srow, sk4) holds W[srow + 16 i][64 kc + sk4 ..]
    // as wreg[16 kc + i].  Both the forward tiles (all n x 64 k) and the backward tiles (64 n x all k) are staged to
    // LDS from it, so the matrix crosses the memory system once per workgroup and no stage waits on a load.
    f32x4 wreg[64];
    const int srow = tid >> 4, sk4 = (tid & 15) * 4;
#pragma unroll
    for (int kc = 0; kc < 4; ++kc)
#pragma unroll
        for (int i = 0; i < 16; ++i)
            wreg[kc * 16 + i] = *reinterpret_cast<const f32x4 *>(Wdv + (size_t)(srow + 16 * i) * NBH + kc * 64 + sk4);
    const float *__restrict__ Wcdv = P + g.p_Wcdv;
    float wc0[4], wc1[4];                          // output layer of the video discriminator, this lane's channels
#pragma unroll
    for (int q = 0; q < 4; ++q) { wc0[q] = Wcdv[q * 64 + lane]; wc1[q] = Wcdv[NBH + q * 64 + lane]; }
    const float wct0 = Wcdv[tid], wct1 = Wcdv[NBH + tid];
    const float bcdv0 = P[g.p_bcdv], bcdv1 = P[g.p_bcdv + 1];
    // (a SCALAR load: as a vector load at the end of the burst above, its consumer `lane == label` - which the compiler hoists up here to
    // keep the result as a lane mask - waited for the whole 256 KB weight burst with s_waitcnt vmcnt(0) before stage A could start)
    const int label = (have && b < g.Bs) ? labels[__builtin_amdgcn_readfirstlane(b)] : -1;
    // small operands of the later stages, requested now as well (each used to cost its stage an exposed round trip):
    // the class bias of this thread's class, the video-discriminator bias of its channel, and for the first relation this
    // wave handles the tuple range and the output-layer bias
    const float bcv_c = (tid >> 2) < C ? P[g.p_bcv + (tid >> 2)] : 0.f;
    const float bdv_n = P[g.p_bdv + (tid >> 4) + 16 * (tid & 15)];

    STAMP(0);
    // ---- A: relation logits, attention, R, V, Vd (WPV waves per video, relations dealt round-robin) ----
    {
        float vacc[4] = {0.f, 0.f, 0.f, 0.f};
        if (have) {
#pragma unroll 1
            for (int j = sub; j < NR; j += WPV) {
                RelIn rel_nxt = rel_cur;
                if (PIPE && j + WPV < NR) load_rel(j + WPV, rel_nxt);      // (wave-uniform) in flight while this relation is reduced
                float d0 = 0.f, d1 = 0.f;
                float r[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    d0 = fmaf(rel_cur.hr[q], rel_cur.w0[q], d0);
                    d1 = fmaf(rel_cur.hr[q], rel_cur.w1[q], d1);
                }
#pragma unroll
                for (int tt = 0; tt < 3; ++tt)     // (tuples past the relation's range were loaded as +0: r + 0 = r exactly, r being a sum of
#pragma unroll                                     // ReLU outputs - the same values as adding only the tuples in range)
                    for (int q = 0; q < 4; ++q) r[q] += rel_cur.zr[tt][q];
                d0 = wave_allreduce_sum(d0) + rel_cur.b0;
                d1 = wave_allreduce_sum(d1) + rel_cur.b1;
                if (PIPE) rel_cur = rel_nxt;
                float w = 0.f;
                if (attn_on) w = 1.f - soft2(d0, d1).H;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    ws[g.o_R + ((size_t)b * NR + j) * NBH + q * 64 + lane] = r[q];
                    vacc[q] += attn_on ? (w + 1.f) * r[q] : r[q];
                }
                if (lane == 0) {
                    ws[g.o_Pr + ((size_t)b * NR + j) * 2 + 0] = d0;
                    ws[g.o_Pr + ((size_t)b * NR + j) * 2 + 1] = d1;
                    ws[g.o_attn + (size_t)b * NR + j] = attn_on ? w : r[0];   // models.py:647-648
                    smem[S_PR + (vloc * 64 + j) * 2 + 0] = d0;
                    smem[S_PR + (vloc * 64 + j) * 2 + 1] = d1;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) smem[S_VPART + wv * NBH + q * 64 + lane] = vacc[q];
    }
    __syncthreads();
#pragma unroll
    for (int v = 0; v < VPW; ++v) {                // thread t <-> channel t: add the waves' partial sums in a fixed order
        float val = 0.f;
#pragma unroll
        for (int s_ = 0; s_ < WPV; ++s_) val += smem[S_VPART + (v * WPV + s_) * NBH + tid];
        float vd = val;
        if (drop_v) vd = val * keep_mask(hs.seed_v, (uint32_t)((b0 + v) * NBH + tid), hs.p_drop_v) * inv_keep_v;
        if (v < nv) {
            ws[g.o_V + (size_t)(b0 + v) * NBH + tid] = val;
            ws[g.o_Vd + (size_t)(b0 + v) * NBH + tid] = vd;
        }
        smem[S_VD + v * NBH + tid] = v < nv ? vd : 0.f;
    }
    STAMP(1);
    // classifier weights -> LDS [C][TROW]
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int f = i * 256 + tid;               // float4 index: row c = f / 64, k4 = (f % 64) * 4
        if (f < C * 64) *reinterpret_cast<f32x4 *>(&smem[S_W + (f >> 6) * TROW + (f & 63) * 4]) = cw[i];
    }
    __syncthreads();

    STAMP(2);
    // ---- B: class logits: 4 threads per class, each a quarter of K; combined on the DPP quad network; one video after the other ----
    {
        const int c = tid >> 2, part = tid & 3;
#pragma unroll
        for (int v = 0; v < VPW; ++v) {
            float acc = 0.f;
            if (c < C) {
                const float *wr = &smem[S_W + c * TROW + part * 64];
                const float *vd = &smem[S_VD + v * NBH + part * 64];
#pragma unroll
                for (int k4 = 0; k4 < 64; k4 += 4) {
                    const float4 w4 = *reinterpret_cast<const float4 *>(wr + k4);
                    const float4 x4 = *reinterpret_cast<const float4 *>(vd + k4);
                    acc = fmaf(w4.x, x4.x, acc); acc = fmaf(w4.y, x4.y, acc); acc = fmaf(w4.z, x4.z, acc); acc = fmaf(w4.w, x4.w, acc);
                }
            }
            acc += dpp_move<0xB1, 0xF>(0.f, acc);      // quad_perm [1,0,3,2]
            acc += dpp_move<0x4E, 0xF>(0.f, acc);      // quad_perm [2,3,0,1]: every lane of the quad holds the class logit
            if (c < C && part == 0) {
                const float yc = acc + bcv_c;
                smem[S_Y + v * 64 + c] = yc;
                if (v < nv) ws[g.o_Y + (size_t)(b0 + v) * C + c] = yc;
            }
        }
    }
    __syncthreads();   // logits visible; done with the classifier tile
    const float y = (lane < C) ? smem[S_Y + vloc * 64 + lane] : -INFINITY;

    STAMP(3);
    // ---- C: Hv = relu(Wdv Vd + bdv) straight from the register copy of Wdv, for the workgroup's VPW videos ----
    // Thread (srow, c16) holds W[srow + 16 i][64 kc + 4 c16 .. +3]: 16 partial dot products over its 16 k, then a sum over the
    // 16 lanes of its DPP row (they share srow and cover all 256 k).  Lane c16 keeps output channel srow + 16 c16.
    {
        const int c16 = tid & 15;
        const int n = srow + 16 * c16;
#pragma unroll
        for (int v = 0; v < VPW; ++v) {
            f32x4 xk[4];
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) xk[kc] = *reinterpret_cast<const f32x4 *>(&smem[S_VD + v * NBH + kc * 64 + sk4]);
            float hv = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float p = 0.f;
#pragma unroll
                for (int kc = 0; kc < 4; ++kc) {
                    const f32x4 w4 = wreg[kc * 16 + i];
                    p = fmaf(w4.x, xk[kc].x, p); p = fmaf(w4.y, xk[kc].y, p); p = fmaf(w4.z, xk[kc].z, p); p = fmaf(w4.w, xk[kc].w, p);
                }
                p += dpp_move<0xB1, 0xF>(0.f, p);      // quad_perm [1,0,3,2]
                p += dpp_move<0x4E, 0xF>(0.f, p);      // quad_perm [2,3,0,1]
                p += dpp_move<0x141, 0xF>(0.f, p);     // row_half_mirror
                p += dpp_move<0x140, 0xF>(0.f, p);     // row_mirror: all 16 lanes of the row hold the sum
                if (c16 == i) hv = p;
            }
            const float h = fmaxf(hv + bdv_n, 0.f);
            smem[S_HV + v * NBH + n] = h;
            if (v < nv) ws[g.o_Hv + (size_t)(b0 + v) * NBH + n] = h;
        }
    }
    __syncthreads();

    STAMP(4);
    // ---- D: video domain logits, losses, gY, gPv (the video's lead wave, lane = class) ----
    {
        float d0 = 0.f, d1 = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float h = smem[S_HV + vloc * NBH + q * 64 + lane];
            d0 = fmaf(h, wc0[q], d0);
            d1 = fmaf(h, wc1[q], d1);
        }
        d0 = wave_allreduce_sum(d0) + bcdv0;
        d1 = wave_allreduce_sum(d1) + bcdv1;
        // class softmax over lanes (lanes >= C hold -inf)
        const float m = wave_allreduce_max(y);
        const float e = lane < C ? expf(y - m) : 0.f;
        const float ls = logf(wave_allreduce_sum(e));
        const float lp = lane < C ? y - m - ls : 0.f;
        const float pr = lane < C ? expf(lp) : 0.f;
        const float Hc = wave_allreduce_sum(-pr * lp);
        float gy = 0.f, g0 = 0.f, g1 = 0.f;
        if (have && lead) {
            const bool is_src = b < g.Bs;
            const bool valid = video_valid(g.Bs, hs.valid_source, hs.valid_target, b);
            const bool cls_on = is_src && valid;
            const Soft2 s = soft2(d0, d1);
            const bool ent_on = (g.flags & TA3N_FLAG_ATTN_ENTROPY) && valid;
            const float ce = hs.gamma * hs.inv_n_ent;
            if (cls_on) {                                                              // main.py:446
                gy = (pr - (lane == label ? 1.f : 0.f)) * hs.inv_n_cls;
                if (lane == label) l_cls = -lp * hs.inv_n_cls;
            }
            if (ent_on) {                                                              // loss.py:20-24
                gy += ce * (1.f + s.H) * (-pr * (lp + Hc));                            // dH/dz_i = -p_i (log p_i + H)
                if (lane == 0) l_ent = (1.f + s.H) * Hc * hs.inv_n_ent;
            }
            if (lane >= C) gy = 0.f;
            if ((g.flags & TA3N_FLAG_ADV_VIDEO) && valid) {                           // main.py:508-538, l = 1
                const int d = is_src ? 0 : 1;
                if (lane == 0) l_vid = -(d ? s.lp1 : s.lp0) * hs.inv_n_vid;
                g0 = (s.p0 - (d == 0 ? 1.f : 0.f)) * hs.inv_n_vid;
                g1 = (s.p1 - (d == 1 ? 1.f : 0.f)) * hs.inv_n_vid;
            }
            if (ent_on) {
                g0 += ce * Hc * (-s.p0 * (s.lp0 + s.H));
                g1 += ce * Hc * (-s.p1 * (s.lp1 + s.H));
            }
            if (lane < C) ws[g.o_gY + (size_t)b * C + lane] = gy;
            if (lane == 0) {
                ws[g.o_Pv + (size_t)b * 2] = d0; ws[g.o_Pv + (size_t)b * 2 + 1] = d1;
                ws[g.o_gPv + (size_t)b * 2] = g0; ws[g.o_gPv + (size_t)b * 2 + 1] = g1;
            }
        }
        if (lead) {
            smem[S_GY + vloc * 64 + lane] = gy;
            if (lane == 0) { smem[S_GPV + vloc * 2] = g0; smem[S_GPV + vloc * 2 + 1] = g1; }
        }
        l_cls = wave_allreduce_sum(l_cls);   // it sits in the label's lane
    }
    __syncthreads();

    STAMP(5);
    // ---- E: gHv = (gPv Wcdv) * [Hv > 0] ----
#pragma unroll
    for (int v = 0; v < VPW; ++v) {
        const float gh = smem[S_HV + v * NBH + tid] > 0.f ? smem[S_GPV + v * 2] * wct0 + smem[S_GPV + v * 2 + 1] * wct1 : 0.f;
        smem[S_GHV + v * NBH + tid] = gh;
        if (v < nv) ws[g.o_gHv + (size_t)(b0 + v) * NBH + tid] = gh;
    }
    __syncthreads();

    // Stage G's operands for this wave's first relation (output-layer rows, Hr, R - none depends on stages B-F) are requested HERE, in
    // front of stage F's arithmetic, and inside G the next relation's while the current one is reduced (round 5: G was 5.3 k cycles for
    // two dependent relations at 9 segments, tools/heads_timing.py).
    struct RelBack { float w20[4], w21[4], hrv[4], rv[4]; };
    auto load_back = [&](int j, RelBack &o) {
        const size_t bj = (size_t)b * NR + j;
        const float *__restrict__ W2 = P + g.p_W2_0 + (size_t)j * g.p_W2_stride;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = q * 64 + lane;
            o.w20[q] = W2[c]; o.w21[q] = W2[NBH + c];
            o.hrv[q] = wsr[g.o_Hr + bj * NBH + c];
            o.rv[q] = ws[g.o_R + bj * NBH + c];      // (R: written by THIS thread in stage A)
        }
    };
    RelBack back_cur;
#pragma unroll
    for (int q = 0; q < 4; ++q) back_cur.w20[q] = back_cur.w21[q] = back_cur.hrv[q] = back_cur.rv[q] = 0.f;
    if (PIPE && have && sub < NR) load_back(sub, back_cur);

    // ---- F: gVt = drop_v'( -beta1 * gHv Wdv + gY Wcv ) from the register copy of Wdv, one video after the other ----
    // Thread (srow, c16) multiplies its 16 rows n = srow + 16 i into partial sums for its 16 input channels
    // k = 64 kc + 4 c16 + e; the 16 threads that share c16 (one per srow) are added through LDS, thread t <-> channel t.
#pragma unroll
    for (int v = 0; v < VPW; ++v) {
        float gh[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) gh[i] = smem[S_GHV + v * NBH + srow + 16 * i];
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            f32x4 p = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const f32x4 w4 = wreg[kc * 16 + i];
                p.x = fmaf(gh[i], w4.x, p.x); p.y = fmaf(gh[i], w4.y, p.y); p.z = fmaf(gh[i], w4.z, p.z); p.w = fmaf(gh[i], w4.w, p.w);
            }
            *reinterpret_cast<f32x4 *>(&smem[S_FPART + srow * NBH + kc * 64 + sk4]) = p;
        }
        __syncthreads();
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc += smem[S_FPART + r * NBH + tid];
        acc *= -hs.beta1;
        // classifier weights column-wise: the [C][TROW] tile staged for stage B is still in place
        for (int c = 0; c < C; ++c) acc = fmaf(smem[S_GY + v * 64 + c], smem[S_W + c * TROW + tid], acc);
        float gv = acc;
        if (drop_v) gv *= keep_mask(hs.seed_v, (uint32_t)((b0 + v) * NBH + tid), hs.p_drop_v);
        gv *= inv_keep_v;
        smem[S_GVT + v * NBH + tid] = gv;
        if (v < nv) ws[g.o_gVt + (size_t)(b0 + v) * NBH + tid] = gv;
        if (VPW > 1) __syncthreads();      // (the partial-sum block is reused by the next video)
    }
    if (VPW == 1) __syncthreads();

    STAMP(6);
    // ---- G: backward of the attention pooling + relation adversarial loss (WPV waves per video) ----
    if (have) {
        const bool is_src = b < g.Bs;
        const bool valid = video_valid(g.Bs, hs.valid_source, hs.valid_target, b);
        const bool adv_rel = (g.flags & TA3N_FLAG_ADV_RELATION) && valid;
        float gv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) gv[q] = smem[S_GVT + vloc * NBH + q * 64 + lane];
#pragma unroll 1
        for (int j = sub; j < NR; j += WPV) {
            const size_t bj = (size_t)b * NR + j;
            if (!PIPE) load_back(j, back_cur);                           // (one relation per wave: requested here, as in round 4)
            RelBack back_nxt = back_cur;
            if (PIPE && j + WPV < NR) load_back(j + WPV, back_nxt);      // (wave-uniform) in flight while this relation is processed
            float w20[4], w21[4], hrv[4], rv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { w20[q] = back_cur.w20[q]; w21[q] = back_cur.w21[q]; hrv[q] = back_cur.hrv[q]; rv[q] = back_cur.rv[q]; }
            if (PIPE) back_cur = back_nxt;
            const float z0 = smem[S_PR + (vloc * 64 + j) * 2], z1 = smem[S_PR + (vloc * 64 + j) * 2 + 1];
            const Soft2 s = soft2(z0, z1);
            float g0 = 0.f, g1 = 0.f;
            if (adv_rel) {                                                             // main.py:508-538, l = 0
                const int d = is_src ? 0 : 1;
                if (lane == 0) l_rel += -(d ? s.lp1 : s.lp0) * hs.inv_n_rel;
                g0 = (s.p0 - (d == 0 ? 1.f : 0.f)) * hs.inv_n_rel;
                g1 = (s.p1 - (d == 1 ? 1.f : 0.f)) * hs.inv_n_rel;
            }
            if (lane == 0) { ws[g.o_gPr + bj * 2] = g0; ws[g.o_gPr + bj * 2 + 1] = g1; }
            float w1 = 1.f;
            if (attn_on) {
                // dL/dw_j = <R_j, dL/dV>;  dw/dz_i = p_i (log p_i + H)   (the weights are not detached, models.py:351-357)
                float dot = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) dot = fmaf(rv[q], gv[q], dot);
                dot = wave_allreduce_sum(dot);
                g0 += dot * s.p0 * (s.lp0 + s.H);
                g1 += dot * s.p1 * (s.lp1 + s.H);
                w1 = 1.f + (1.f - s.H);
            }
            if (lane == 0) { ws[g.o_gPrT + bj * 2] = g0; ws[g.o_gPrT + bj * 2 + 1] = g1; }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = q * 64 + lane;
                ws[g.o_gRa + bj * NBH + c] = w1 * gv[q];
                ws[g.o_gHr + bj * NBH + c] = hrv[q] > 0.f ? g0 * w20[q] + g1 * w21[q] : 0.f;
            }
        }
    }
    {
        auto lane0 = [](float x) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 0)); };
        const float lr = lane0(l_rel), lv = lane0(l_vid), le = lane0(l_ent);
        if (lane < 8) {
            float v = 0.f;
            if (lane == 1) v = l_cls;
            if (lane == 2) v = lr;
            if (lane == 3) v = lv;
            if (lane == 5) v = le;
            smem[S_LOSS + wv * 8 + lane] = v;
        }
    }
    STAMP(7);
    write_loss_part(smem, ws, g.o_loss_part, (int)blockIdx.x - g.n_frm_wg, hs.gamma, S_LOSS);
    STAMP(8);
}

// Frame rows: Pf, frame adversarial CE, gPf, gHf = (gPf Wcd) * [Hf > 0], and this workgroup's
// partial sums of dWcd = gPf^T Hf and dbcd.  FQ = ceil(F / 64) channels per lane; each wave keeps
// its 4 rows in registers.
template <int FQ>
__device__ __forceinline__ void frame_wg(const Geom g, const Ptrs ptrs, float *smem, int wg) {
    float *__restrict__ ws = ptrs.ws;
    const float *__restrict__ wsr = ptrs.ws;       // Hf is only read
    const float *__restrict__ P = ptrs.p;
    const Hyper *__restrict__ hy = reinterpret_cast<const Hyper *>(ptrs.ws + g.o_hyper);
    const StepScalars hs = step_scalars(hy);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int F = g.F, T = g.T, BT = g.B * g.T;
    const float *__restrict__ W0 = P + g.p_Wcd, *__restrict__ W1 = W0 + F;
    const float bc0 = P[g.p_bcd], bc1 = P[g.p_bcd + 1];
    const bool adv = (g.flags & TA3N_FLAG_ADV_FRAME) != 0;
    constexpr int NROW = RPW / 4;
    float w0[FQ], w1[FQ], hf[NROW][FQ];
#pragma unroll
    for (int q = 0; q < FQ; ++q) {
        const int k = q * 64 + lane;
        w0[q] = k < F ? W0[k] : 0.f;
        w1[q] = k < F ? W1[k] : 0.f;
    }
    float a0[FQ], a1[FQ], sg0 = 0.f, sg1 = 0.f, l_frm = 0.f;
#pragma unroll
    for (int q = 0; q < FQ; ++q) { a0[q] = 0.f; a1[q] = 0.f; }
    // g.heads_rpw rows per workgroup, in groups of RPW = 16 (4 rows per wave held in registers); the plan picks heads_rpw so
    // that video + frame workgroups all fit on the chip at once (a video workgroup owns a compute unit: the 11th..266th
    // workgroup of a 266-workgroup grid would otherwise start when the first video workgroups END)
#pragma unroll 1
    for (int rg = 0; rg < g.heads_rpw / RPW; ++rg) {
    const int row0 = wg * g.heads_rpw + rg * RPW;
#pragma unroll
    for (int i = 0; i < NROW; ++i) {
        const int r = row0 + wv + 4 * i;
#pragma unroll
        for (int q = 0; q < FQ; ++q) {
            const int k = q * 64 + lane;
            hf[i][q] = (r < BT && k < F) ? wsr[g.o_Hf + (size_t)r * F + k] : 0.f;
        }
    }
    float d0[NROW], d1[NROW];
#pragma unroll
    for (int i = 0; i < NROW; ++i) {
        d0[i] = 0.f; d1[i] = 0.f;
#pragma unroll
        for (int q = 0; q < FQ; ++q) { d0[i] = fmaf(hf[i][q], w0[q], d0[i]); d1[i] = fmaf(hf[i][q], w1[q], d1[i]); }
    }
#pragma unroll
    for (int i = 0; i < NROW; ++i) { d0[i] = wave_allreduce_sum(d0[i]) + bc0; d1[i] = wave_allreduce_sum(d1[i]) + bc1; }
#pragma unroll
    for (int i = 0; i < NROW; ++i) {
        const int r = row0 + wv + 4 * i;
        if (r < BT) {   // wave-uniform
            const int b = r / T;
            const bool valid = video_valid(g.Bs, hs.valid_source, hs.valid_target, b);
            float g0 = 0.f, g1 = 0.f;
            if (adv && valid) {                                                        // main.py:508-538, l = 2
                const Soft2 s = soft2(d0[i], d1[i]);
                const int d = b < g.Bs ? 0 : 1;
                l_frm += -(d ? s.lp1 : s.lp0) * hs.inv_n_frm;
                g0 = (s.p0 - (d == 0 ? 1.f : 0.f)) * hs.inv_n_frm;
                g1 = (s.p1 - (d == 1 ? 1.f : 0.f)) * hs.inv_n_frm;
            }
            if (lane == 0) {
                ws[g.o_Pf + (size_t)r * 2] = d0[i]; ws[g.o_Pf + (size_t)r * 2 + 1] = d1[i];
                ws[g.o_gPf + (size_t)r * 2] = g0; ws[g.o_gPf + (size_t)r * 2 + 1] = g1;
            }
            sg0 += g0; sg1 += g1;
#pragma unroll
            for (int q = 0; q < FQ; ++q) {
                const int k = q * 64 + lane;
                const float h = hf[i][q];
                if (k < F) {
                    const float gh = h > 0.f ? g0 * w0[q] + g1 * w1[q] : 0.f;
                    ws[g.o_gHf + (size_t)r * F + k] = gh;
                    if (g.o_ws16 >= 0)   // bf16 twin (TA3N_FLAG_BF16_STORE): the gradient launch reads it as a GEMM operand
                    {
                        const unsigned hb = pack_bf16(gh, 0.f);
                        unsigned short *tw = reinterpret_cast<unsigned short *>(ws + g.o_ws16) + g.o_gHf + (size_t)r * F + k;
                        *tw = (unsigned short)hb;
                        if (g.pair_delta) tw[2 * (size_t)g.pair_delta] = (unsigned short)pack_bf16_lo(gh, 0.f, hb);      // pair twins: the lo plane
                    }
                }
                a0[q] = fmaf(g0, h, a0[q]);
                a1[q] = fmaf(g1, h, a1[q]);
            }
        }
    }
    }   // row groups
    // cross-wave sums through LDS: [4 waves][2][FQ*64] then [4][2] for the bias partials
    constexpr int FP = FQ * 64;
#pragma unroll
    for (int q = 0; q < FQ; ++q) {
        smem[(wv * 2 + 0) * FP + q * 64 + lane] = a0[q];
        smem[(wv * 2 + 1) * FP + q * 64 + lane] = a1[q];
    }
    if (lane == 0) { smem[8 * FP + wv * 2] = sg0; smem[8 * FP + wv * 2 + 1] = sg1; }
    if (lane < 8) smem[S_LOSS_MIN + wv * 8 + lane] = lane == 4 ? l_frm : 0.f;
    __syncthreads();
    float *__restrict__ part = ws + g.o_fh_part + (size_t)wg * 2 * F;
    for (int i = tid; i < 2 * F; i += 256) {
        const int c = i / F, k = i - c * F;
        part[i] = (smem[(0 + c) * FP + k] + smem[(2 + c) * FP + k]) + (smem[(4 + c) * FP + k] + smem[(6 + c) * FP + k]);
    }
    if (tid < 2)
        ws[g.o_fh_bpart + (size_t)wg * 2 + tid] = (smem[8 * FP + tid] + smem[8 * FP + 2 + tid]) + (smem[8 * FP + 4 + tid] + smem[8 * FP + 6 + tid]);
    write_loss_part(smem, ws, g.o_loss_part, g.n_vid_wg + wg, hs.gamma, S_LOSS_MIN);
}

template <int FQ, int VPW, bool PIPE>
__global__ __launch_bounds__(256) void heads_kernel(Geom g, Ptrs ptrs) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // the (short) frame workgroups come first in the grid: if the grid does not fit on the chip at once, the workgroups that
    // start late are video workgroups behind finished frame workgroups, not frame workgroups behind finished video ones
    if ((int)blockIdx.x < g.n_frm_wg) frame_wg<FQ>(g, ptrs, smem, (int)blockIdx.x);
    else video_wg<VPW, PIPE>(g, ptrs, smem);
}

template <int FQ, int VPW, bool PIPE>
int launch_fq_vpw_pipe(const Geom &g, const Ptrs &ptrs, hipStream_t stream) {
    static_assert(8 * FQ * 64 + 16 <= S_LOSS_MIN, "the frame partials must stay below the loss slots");
    static std::once_flag attr_once;   // > 64 KiB of dynamic LDS needs the opt-in once per process; callers may be DataParallel's
                                       // one-thread-per-replica workers (re-entrancy), hence call_once and no plain flag
    std::call_once(attr_once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(heads_kernel<FQ, VPW, PIPE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    hipLaunchKernelGGL((heads_kernel<FQ, VPW, PIPE>), dim3(g.n_vid_wg + g.n_frm_wg), dim3(256), (size_t)Lds<VPW>::TOTAL * sizeof(float), stream, g, ptrs);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <int FQ, int VPW>
int launch_fq_vpw(const Geom &g, const Ptrs &ptrs, hipStream_t stream) {
    return g.n_rel > 4 / VPW ? launch_fq_vpw_pipe<FQ, VPW, true>(g, ptrs, stream) : launch_fq_vpw_pipe<FQ, VPW, false>(g, ptrs, stream);
}

template <int FQ>
int launch_fq(const Geom &g, const Ptrs &ptrs, hipStream_t stream) {
    if (g.heads_vpw == 2) return launch_fq_vpw<FQ, 2>(g, ptrs, stream);
    return launch_fq_vpw<FQ, 1>(g, ptrs, stream);
}

}  // namespace

namespace ta3n {

bool heads_supported(int NB, int C, int F) { return NB == NBH && C <= 64 && F <= 2048; }

int launch_heads(const Geom &g, const Ptrs &ptrs, hipStream_t stream) {
    if (!heads_supported(g.NB, g.C, g.F)) return -1;
    const int fq = (g.F + 63) / 64;
    if (fq <= 1) return launch_fq<1>(g, ptrs, stream);
    if (fq <= 2) return launch_fq<2>(g, ptrs, stream);
    if (fq <= 4) return launch_fq<4>(g, ptrs, stream);
    if (fq <= 8) return launch_fq<8>(g, ptrs, stream);
    if (fq <= 16) return launch_fq<16>(g, ptrs, stream);
    return launch_fq<32>(g, ptrs, stream);
}

}  // namespace ta3n
