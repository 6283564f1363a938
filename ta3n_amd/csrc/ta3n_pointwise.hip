// Non-GEMM kernels of the TA3N train step (gfx950, wave64).
//
//  pool_fwd   one wavefront per video: relation-discriminator logits (the 2-wide
//             second layer, reference models.py:479), transferable attention
//             w = 1 - H(softmax) (models.py:351-357), R_j = sum of the scale's
//             tuple activations (TRNmodule.py:73-79), V = sum_j (1+w_j) R_j
//             (models.py:379-388, 651), dropout_v (models.py:679).
//  loss       CE + 3 adversarial CE + attentive entropy and all logit gradients
//             (main.py:439-451, 508-538, 559-562; loss.py:15-25).
//  pool_bwd   backward of pool_fwd including the un-detached attention path.
//  grad_norm / sgd   clip_grad_norm_ + Nesterov SGD with weight decay over the
//             flat live-parameter prefix (main.py:578-583).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>

#include "ta3n_kernels.h"
#include "../../include/ta3n_hip.h"

using namespace ta3n;

namespace {

template <int Q>   // Q = NB / 64 channels per lane
__global__ __launch_bounds__(256) void pool_fwd_kernel(Geom g, Ptrs ptrs) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    float *__restrict__ ws = ptrs.ws;
    // The loss kernel accumulates its logging scalars with atomics; they are cleared here
    // (this kernel always runs between two loss kernels).  A hipMemsetAsync node for these
    // 32 bytes replayed garbage under hipGraph on ROCm 7.2, eager launches were fine.
    if (blockIdx.x == 0 && threadIdx.x < 8) ws[g.o_losses + threadIdx.x] = 0.f;
    if (b >= g.B) return;
    const Hyper *__restrict__ hy = reinterpret_cast<const Hyper *>(ws + g.o_hyper);
    const int *__restrict__ tf = reinterpret_cast<const int *>(ws + g.o_tuple_first);
    const int NB = g.NB, NR = g.n_rel, NT = g.n_tuples;
    const bool attn_on = (g.flags & TA3N_FLAG_TRANS_ATTN) != 0;
    float vacc[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) vacc[q] = 0.f;
    for (int j = 0; j < NR; ++j) {
        const float *__restrict__ W2 = ptrs.p + g.p_W2_0 + (size_t)j * g.p_W2_stride;
        const float *__restrict__ b2 = ptrs.p + g.p_b2_0 + (size_t)j * g.p_b2_stride;
        const float *__restrict__ hr = ws + g.o_Hr + ((size_t)b * NR + j) * NB;
        float d0 = 0.f, d1 = 0.f;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int c = q * 64 + lane;
            const float h = hr[c];
            d0 = fmaf(h, W2[c], d0);
            d1 = fmaf(h, W2[NB + c], d1);
        }
        d0 = wave_allreduce_sum(d0) + b2[0];
        d1 = wave_allreduce_sum(d1) + b2[1];
        float w = 0.f;
        if (attn_on) w = 1.f - soft2(d0, d1).H;
        float r0 = 0.f;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int c = q * 64 + lane;
            float r = 0.f;
            for (int t = tf[j]; t < tf[j + 1]; ++t) r += ws[g.o_Zr + ((size_t)b * NT + t) * NB + c];
            ws[g.o_R + ((size_t)b * NR + j) * NB + c] = r;
            vacc[q] += attn_on ? (w + 1.f) * r : r;
            if (q == 0) r0 = r;
        }
        if (lane == 0) {
            ws[g.o_Pr + ((size_t)b * NR + j) * 2 + 0] = d0;
            ws[g.o_Pr + ((size_t)b * NR + j) * 2 + 1] = d1;
            ws[g.o_attn + (size_t)b * NR + j] = attn_on ? w : r0;   // models.py:647-648 returns feat[:,:,0] without attention
        }
    }
    const float inv_keep = hyper_scale(hy, SK_INV_KEEP_V);
    const bool drop = hy->train != 0 && hy->p_drop_v > 0.f;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int c = q * 64 + lane;
        const float v = vacc[q];
        ws[g.o_V + (size_t)b * NB + c] = v;
        float vd = v;
        if (drop) vd = v * keep_mask(hy->seed_v, (uint32_t)(b * NB + c), hy->p_drop_v) * inv_keep;
        ws[g.o_Vd + (size_t)b * NB + c] = vd;
    }
}

template <int Q>
__global__ __launch_bounds__(256) void pool_bwd_kernel(Geom g, Ptrs ptrs) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (b >= g.B) return;
    float *__restrict__ ws = ptrs.ws;
    const int NB = g.NB, NR = g.n_rel;
    const bool attn_on = (g.flags & TA3N_FLAG_TRANS_ATTN) != 0;
    float gv[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) gv[q] = ws[g.o_gVt + (size_t)b * NB + q * 64 + lane];
    if (g.o_gV_ext > 0) {   // TA3N_FLAG_FEATURE_GRADS: the caller's gradient at the pooled feature (discrepancy losses on feat[1])
#pragma unroll
        for (int q = 0; q < Q; ++q) gv[q] += ws[g.o_gV_ext + (size_t)b * NB + q * 64 + lane];
    }
    for (int j = 0; j < NR; ++j) {
        const size_t bj = (size_t)b * NR + j;
        const float *__restrict__ W2 = ptrs.p + g.p_W2_0 + (size_t)j * g.p_W2_stride;
        float g0 = ws[g.o_gPr + bj * 2 + 0], g1 = ws[g.o_gPr + bj * 2 + 1];
        float w1 = 1.f;
        if (attn_on) {
            // dL/dw_j = <R_j, dL/dV> (+ upstream gradient on the attention output);
            // dw/dz_i = p_i (log p_i + H)   (the weights are not detached, models.py:351-357)
            float dot = 0.f;
#pragma unroll
            for (int q = 0; q < Q; ++q) dot = fmaf(ws[g.o_R + bj * NB + q * 64 + lane], gv[q], dot);
            dot = wave_allreduce_sum(dot) + ws[g.o_gattn + bj];
            const Soft2 s = soft2(ws[g.o_Pr + bj * 2 + 0], ws[g.o_Pr + bj * 2 + 1]);
            g0 += dot * s.p0 * (s.lp0 + s.H);
            g1 += dot * s.p1 * (s.lp1 + s.H);
            w1 = 1.f + (1.f - s.H);
        }
        if (lane == 0) {
            ws[g.o_gPrT + bj * 2 + 0] = g0;
            ws[g.o_gPrT + bj * 2 + 1] = g1;
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int c = q * 64 + lane;
            ws[g.o_gRa + bj * NB + c] = w1 * gv[q];
            const float gh = g0 * W2[c] + g1 * W2[NB + c];
            ws[g.o_gHr + bj * NB + c] = ws[g.o_Hr + bj * NB + c] > 0.f ? gh : 0.f;
        }
    }
}

__device__ __forceinline__ float block_sum(float v, float *red) {
    v = wave_allreduce_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
    return t;
}

// One thread per logit row: video rows [0,B), relation rows, frame rows.
__global__ __launch_bounds__(256) void loss_kernel(Geom g, Ptrs ptrs) {
    __shared__ float red[8];
    float *__restrict__ ws = ptrs.ws;
    const Hyper *__restrict__ hy = reinterpret_cast<const Hyper *>(ws + g.o_hyper);
    const int *__restrict__ labels = reinterpret_cast<const int *>(ws + g.o_labels);
    const int B = g.B, NR = g.n_rel, T = g.T, C = g.C;
    const int n_vid = B, n_rel = B * NR, n_frm = B * T;
    const int rid = blockIdx.x * blockDim.x + threadIdx.x;
    float l_cls = 0.f, l_rel = 0.f, l_vid = 0.f, l_frm = 0.f, l_ent = 0.f;
    if (rid < n_vid) {
        const int b = rid;
        const bool is_src = b < g.Bs;
        const bool valid = is_src ? (b < hy->valid_source) : (b - g.Bs < hy->valid_target);
        const float *__restrict__ y = ws + g.o_Y + (size_t)b * C;
        float *__restrict__ gy = ws + g.o_gY + (size_t)b * C;
        float *__restrict__ gpv = ws + g.o_gPv + (size_t)b * 2;
        float m = y[0];
        for (int i = 1; i < C; ++i) m = fmaxf(m, y[i]);
        float sum = 0.f;
        for (int i = 0; i < C; ++i) sum += expf(y[i] - m);
        const float ls = logf(sum);
        float Hc = 0.f;
        for (int i = 0; i < C; ++i) {
            const float lp = y[i] - m - ls;
            Hc -= expf(lp) * lp;
        }
        const bool cls_on = is_src && valid;
        const int lab = cls_on ? labels[b] : -1;
        if (cls_on) l_cls = -(y[lab] - m - ls) * hy->inv_n_cls;                      // main.py:446
        const float z0 = ws[g.o_Pv + (size_t)b * 2], z1 = ws[g.o_Pv + (size_t)b * 2 + 1];
        const Soft2 s = soft2(z0, z1);
        const bool ent_on = (g.flags & TA3N_FLAG_ATTN_ENTROPY) && valid;
        const float ce = hy->gamma * hy->inv_n_ent;
        if (ent_on) l_ent = (1.f + s.H) * Hc * hy->inv_n_ent;                         // loss.py:20-24
        for (int i = 0; i < C; ++i) {
            const float lp = y[i] - m - ls;
            const float p = expf(lp);
            float gi = 0.f;
            if (cls_on) gi = (p - (i == lab ? 1.f : 0.f)) * hy->inv_n_cls;
            if (ent_on) gi += ce * (1.f + s.H) * (-p * (lp + Hc));                    // dH/dz_i = -p_i (log p_i + H)
            gy[i] = gi;
        }
        float g0 = 0.f, g1 = 0.f;
        // Without relation features (avgpool) the reference's relation slot of pred_domain holds the VIDEO logits once more
        // ("add dummy tensors", models.py:707-708), so place_adv[0] = 'Y' adds the video-level CE a second time.
        const float vmult = ((g.flags & TA3N_FLAG_ADV_VIDEO) ? 1.f : 0.f) + ((NR == 0 && (g.flags & TA3N_FLAG_ADV_RELATION)) ? 1.f : 0.f);
        if (vmult > 0.f && valid) {                                                  // main.py:508-538, l = 1 (and l = 0 for avgpool)
            const int d = is_src ? 0 : 1;
            l_vid = -(d ? s.lp1 : s.lp0) * hy->inv_n_vid * vmult;
            g0 = (s.p0 - (d == 0 ? 1.f : 0.f)) * hy->inv_n_vid * vmult;
            g1 = (s.p1 - (d == 1 ? 1.f : 0.f)) * hy->inv_n_vid * vmult;
        }
        if (ent_on) {
            g0 += ce * Hc * (-s.p0 * (s.lp0 + s.H));
            g1 += ce * Hc * (-s.p1 * (s.lp1 + s.H));
        }
        gpv[0] = g0; gpv[1] = g1;
    } else if (rid < n_vid + n_rel + n_frm) {
        const bool is_rel = rid < n_vid + n_rel;
        const int row = is_rel ? rid - n_vid : rid - n_vid - n_rel;
        const int b = is_rel ? row / NR : row / T;
        const bool is_src = b < g.Bs;
        const bool valid = is_src ? (b < hy->valid_source) : (b - g.Bs < hy->valid_target);
        const float *__restrict__ z = ws + (is_rel ? g.o_Pr : g.o_Pf) + (size_t)row * 2;
        float *__restrict__ gz = ws + (is_rel ? g.o_gPr : g.o_gPf) + (size_t)row * 2;
        const bool on = valid && (g.flags & (is_rel ? TA3N_FLAG_ADV_RELATION : TA3N_FLAG_ADV_FRAME));
        float g0 = 0.f, g1 = 0.f;
        if (on) {
            const float inv_n = is_rel ? hy->inv_n_rel : hy->inv_n_frm;
            const Soft2 s = soft2(z[0], z[1]);
            const int d = is_src ? 0 : 1;
            const float l = -(d ? s.lp1 : s.lp0) * inv_n;
            if (is_rel) l_rel = l; else l_frm = l;
            g0 = (s.p0 - (d == 0 ? 1.f : 0.f)) * inv_n;
            g1 = (s.p1 - (d == 1 ? 1.f : 0.f)) * inv_n;
        }
        gz[0] = g0; gz[1] = g1;
    }
    // loss scalars are for logging only; gradients above are exact per row
    const float s_cls = block_sum(l_cls, red), s_rel = block_sum(l_rel, red), s_vid = block_sum(l_vid, red);
    const float s_frm = block_sum(l_frm, red), s_ent = block_sum(l_ent, red);
    if (threadIdx.x == 0) {
        float *L = ws + g.o_losses;
        const float total = s_cls + s_rel + s_vid + s_frm + hy->gamma * s_ent;
        if (total != 0.f) atomicAdd(L + 0, total);
        if (s_cls != 0.f) atomicAdd(L + 1, s_cls);
        if (s_rel != 0.f) atomicAdd(L + 2, s_rel);
        if (s_vid != 0.f) atomicAdd(L + 3, s_vid);
        if (s_frm != 0.f) atomicAdd(L + 4, s_frm);
        if (s_ent != 0.f) atomicAdd(L + 5, s_ent);
    }
}

__global__ __launch_bounds__(256) void grad_norm_kernel(const float *__restrict__ grads, float *__restrict__ part, int n4) {
    __shared__ float red[8];
    float acc = 0.f;
    const float4 *__restrict__ g4 = reinterpret_cast<const float4 *>(grads);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
        const float4 v = g4[i];
        acc = fmaf(v.x, v.x, acc); acc = fmaf(v.y, v.y, acc); acc = fmaf(v.z, v.z, acc); acc = fmaf(v.w, v.w, acc);
    }
    const float s = block_sum(acc, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// clip_grad_norm_ (total_norm over all grads, coef = clip/(norm+1e-6) clamped to 1)
// then torch.optim.SGD(nesterov=True): g += wd*p; buf = mu*buf + g; g += mu*buf; p -= lr*g.
// norm_off / norm_n: the partial sums of squares to add up (grad_norm_kernel's per-block partials, or the
// per-tile partials the fused step's gradient tiles left behind).
__global__ __launch_bounds__(256) void sgd_kernel(Geom g, float *__restrict__ params, const float *__restrict__ grads,
                                                  float *__restrict__ mom, float *__restrict__ ws, int n4, int norm_off, int norm_n) {
    __shared__ float red[8];
    const Hyper *__restrict__ hy = reinterpret_cast<const Hyper *>(ws + g.o_hyper);
    float4 *__restrict__ p4 = reinterpret_cast<float4 *>(params);
    float4 *__restrict__ m4 = reinterpret_cast<float4 *>(mom);
    const float4 *__restrict__ g4 = reinterpret_cast<const float4 *>(grads);
    // the first elements are requested before the norm partials are added up: every workgroup repeats that reduction
    // (a fixed-order sum of a few thousand floats), and the stream should not wait behind it
    const int stride = gridDim.x * blockDim.x;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f), m = p, gr = p;
    if (i < n4) { p = p4[i]; m = m4[i]; gr = g4[i]; }
    float acc = 0.f;
    acc = strided_partial_sum(ws + norm_off, norm_n, (int)threadIdx.x, (int)blockDim.x);
    const float total = sqrtf(block_sum(acc, red));
    float coef = 1.f;
    if (hy->clip > 0.f) coef = fminf(hy->clip / (total + 1e-6f), 1.f);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ws[g.o_grad_norm] = total;
        ws[g.o_grad_norm + 1] = coef;
    }
    const float lr = hy->lr, mu = hy->momentum, wd = hy->weight_decay;
    while (i < n4) {
        const int nxt = i + stride;
        float4 pn = p, mn = m, gn = gr;
        if (nxt < n4) { pn = p4[nxt]; mn = m4[nxt]; gn = g4[nxt]; }   // next element in flight while this one is updated
        float gg[4] = {gr.x, gr.y, gr.z, gr.w};
        float pp[4] = {p.x, p.y, p.z, p.w};
        float mm[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float d = fmaf(wd, pp[e], gg[e] * coef);
            mm[e] = fmaf(mu, mm[e], d);
            d = fmaf(mu, mm[e], d);
            pp[e] = fmaf(-lr, d, pp[e]);
        }
        p4[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
        m4[i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
        if (g.o_p16 >= 0) {
            const unsigned h0 = pack_bf16(pp[0], pp[1]), h1 = pack_bf16(pp[2], pp[3]);
            reinterpret_cast<uint2 *>(ws + g.o_p16)[i] = make_uint2(h0, h1);
            if (g.pair_delta) reinterpret_cast<uint2 *>(ws + g.o_p16 + g.pair_delta)[i] = make_uint2(pack_bf16_lo(pp[0], pp[1], h0), pack_bf16_lo(pp[2], pp[3], h1));
        }
        i = nxt; p = pn; m = mn; gr = gn;
    }
}


// TA3N_AGG_AVGPOOL (BASELINE configs[0]): everything between F1 and gZ1 for one video per workgroup.
//   V = mean_t F1[b,t,:]  (models.py:421-433, AvgPool2d over the segments);  Vd = dropout_v(V)  (:679);  Y = Wcv Vd + bcv  (:686)
//   loss = CE(Y[source rows], label) / n_source  (main.py:446);  gY = (softmax - onehot) / n_source on valid source rows
//   gVd = Wcv^T gY;  gV = dropout_v'(gVd);  gZ1[b,t,:] = gV / T * [F1[b,t,:] > 0] / keep_i   (dropout_i and ReLU of the frame FC)
// Target rows are forwarded (Y is an output) and get zero gradients, as in the reference's source-only configuration.
__global__ __launch_bounds__(256) void pool_cls_kernel(Geom g, Ptrs ptrs) {
    extern __shared__ float sm[];                 // [F] Vd, then [F] mask_v/keep_v, [64] Y / gY
    float *__restrict__ s_vd = sm, *__restrict__ s_mk = sm + g.F, *__restrict__ s_y = sm + 2 * g.F;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int F = g.F, C = g.C, T = g.T;
    float *__restrict__ ws = ptrs.ws;
    const float *__restrict__ P = ptrs.p;
    const Hyper *__restrict__ hy = reinterpret_cast<const Hyper *>(ws + g.o_hyper);
    const bool train = hy->train != 0;
    const bool drop_v = train && hy->p_drop_v > 0.f;
    const float inv_keep_v = hyper_scale(hy, SK_INV_KEEP_V), inv_keep_i = hyper_scale(hy, SK_INV_KEEP_I);
    const float inv_T = 1.f / (float)T;
    for (int k = tid; k < F; k += 256) {
        float v = 0.f;
        for (int t = 0; t < T; ++t) v += ws[g.o_F1 + ((size_t)b * T + t) * F + k];
        v *= inv_T;
        const float mk = drop_v ? keep_mask(hy->seed_v, (uint32_t)(b * F + k), hy->p_drop_v) * inv_keep_v : 1.f;
        ws[g.o_V + (size_t)b * F + k] = v;
        ws[g.o_Vd + (size_t)b * F + k] = v * mk;
        s_vd[k] = v * mk;
        s_mk[k] = mk;
    }
    __syncthreads();
    for (int c = wave; c < C; c += 4) {           // one wave per class: dot(Wcv[c,:], Vd)
        float acc = 0.f;
        for (int k = lane; k < F; k += 64) acc = fmaf(P[g.p_Wcv + (size_t)c * F + k], s_vd[k], acc);
        acc = wave_allreduce_sum(acc);
        if (lane == 0) {
            const float y = acc + P[g.p_bcv + c];
            s_y[c] = y;
            ws[g.o_Y + (size_t)b * C + c] = y;
        }
    }
    __syncthreads();
    if (wave == 0) {                              // softmax cross-entropy on the C <= 64 logits
        const bool on = b < g.Bs && b < hy->valid_source;
        const int label = on ? reinterpret_cast<const int32_t *>(ws + g.o_labels)[b] : -1;
        const float y = lane < C ? s_y[lane] : -INFINITY;
        const float m = wave_allreduce_max(y);
        const float e = lane < C ? expf(y - m) : 0.f;
        const float ls = logf(wave_allreduce_sum(e));
        const float lp = lane < C ? y - m - ls : 0.f;
        const float gy = (on && lane < C) ? (expf(lp) - (lane == label ? 1.f : 0.f)) * hy->inv_n_cls : 0.f;
        const float l = wave_allreduce_sum((on && lane == label) ? -lp * hy->inv_n_cls : 0.f);
        if (lane < C) ws[g.o_gY + (size_t)b * C + lane] = gy;
        if (lane < 8) ws[g.o_loss_part + (size_t)b * 8 + lane] = lane < 2 ? l : 0.f;   // {loss, loss_c, 0 ...}: losses region layout
    }
    __syncthreads();                               // s_y is free again: reuse it for gY
    if (tid < 64) s_y[tid] = tid < C ? ws[g.o_gY + (size_t)b * C + tid] : 0.f;
    __syncthreads();
    unsigned short *__restrict__ twin = g.o_ws16 >= 0 ? reinterpret_cast<unsigned short *>(ws + g.o_ws16) : nullptr;
    for (int k = tid; k < F; k += 256) {
        float gvd = 0.f;
        for (int c = 0; c < C; ++c) gvd = fmaf(P[g.p_Wcv + (size_t)c * F + k], s_y[c], gvd);
        const float gf = gvd * s_mk[k] * inv_T * (train ? inv_keep_i : 1.f);
        for (int t = 0; t < T; ++t) {
            const size_t idx = ((size_t)b * T + t) * F + k;
            const float gz = ws[g.o_F1 + idx] > 0.f ? gf : 0.f;
            ws[g.o_gZ1 + idx] = gz;
            if (twin) {
                const unsigned h = pack_bf16(gz, 0.f);
                twin[g.o_gZ1 + idx] = (unsigned short)h;
                if (g.pair_delta) twin[2 * (size_t)g.pair_delta + g.o_gZ1 + idx] = (unsigned short)pack_bf16_lo(gz, 0.f, h);
            }
        }
    }
}

// TA3N_AGG_AVGPOOL, general (TemPooling with or without the adversarial branches): aggregate_frames "1. averaging"
// (models.py:421-433: AvgPool2d over the segments) and dropout_v (:679).  One workgroup per video.
__global__ __launch_bounds__(256) void pool_avg_fwd_kernel(Geom g, Ptrs ptrs) {
    float *__restrict__ ws = ptrs.ws;
    const Hyper *__restrict__ hy = reinterpret_cast<const Hyper *>(ws + g.o_hyper);
    const int b = blockIdx.x, F = g.F, T = g.T;
    const bool drop_v = hy->train != 0 && hy->p_drop_v > 0.f;
    const float inv_keep_v = hyper_scale(hy, SK_INV_KEEP_V), inv_T = 1.f / (float)T;
    if (blockIdx.x == 0 && threadIdx.x < 8) ws[g.o_losses + threadIdx.x] = 0.f;   // the loss kernel accumulates into them
    for (int k = threadIdx.x; k < F; k += 256) {
        float v = 0.f;
        for (int t = 0; t < T; ++t) v += ws[g.o_F1 + ((size_t)b * T + t) * F + k];
        v *= inv_T;
        ws[g.o_V + (size_t)b * F + k] = v;
        ws[g.o_Vd + (size_t)b * F + k] = drop_v ? v * keep_mask(hy->seed_v, (uint32_t)(b * F + k), hy->p_drop_v) * inv_keep_v : v;
    }
}

// Backward of the averaging: every segment of video b receives gVt[b] / T (gVt already carries dropout_v's mask and scale,
// it is the epilogue of the launch that made it).  With a live frame discriminator that is the additive operand ("gRa"
// region) of the launch that forms gZ1 = (-beta2 gHf Wfd + base) * [F1 > 0] / keep_i; without one it IS gZ1 after the mask.
__global__ __launch_bounds__(256) void pool_avg_bwd_kernel(Geom g, Ptrs ptrs, int direct) {
    float *__restrict__ ws = ptrs.ws;
    const Hyper *__restrict__ hy = reinterpret_cast<const Hyper *>(ws + g.o_hyper);
    const int row = blockIdx.x, b = row / g.T, F = g.F;
    const float inv_T = 1.f / (float)g.T, inv_keep_i = hyper_scale(hy, SK_INV_KEEP_I);
    for (int k = threadIdx.x; k < F; k += 256) {
        float gv = ws[g.o_gVt + (size_t)b * F + k];
        if (g.o_gV_ext > 0) gv += ws[g.o_gV_ext + (size_t)b * F + k];      // TA3N_FLAG_FEATURE_GRADS: the caller's gradient at V (feat[1])
        const float base = gv * inv_T;
        if (direct) ws[g.o_gZ1 + (size_t)row * F + k] = ws[g.o_F1 + (size_t)row * F + k] > 0.f ? base * inv_keep_i : 0.f;
        else ws[g.o_gRa + (size_t)row * F + k] = base;
    }
}

// TSNDataSet.__getitem__ for a whole batch on the device (reference dataset.py:103-116, 128-144, new_length 1):
// one workgroup per output row (video v, segment x).  The segment index is computed in float64 exactly as the
// reference's Python does (tick = n / T; int(tick / 2.0 + tick * x)); clips shorter than T repeat their last frame.
__device__ __forceinline__ void gather_row_f32(int row, const float *__restrict__ store, const int64_t *__restrict__ first_row,
                                               const int32_t *__restrict__ num_frames, const int32_t *__restrict__ labels,
                                               const int32_t *__restrict__ video_ids, int T, int D,
                                               float *__restrict__ out, int32_t *__restrict__ labels_out,
                                               int32_t *__restrict__ seg_out, uint2 *__restrict__ out16, int64_t pair_delta) {
    const int v = row / T, x = row - v * T;
    const int vid = video_ids[v];
    const int nf = num_frames[vid];
    int off;
    if (nf >= T) {
        const double tick = (double)nf / (double)T;
        off = (int)(tick / 2.0 + tick * (double)x);
    } else {
        off = x < nf ? x : nf - 1;
    }
    if (threadIdx.x == 0) {
        if (seg_out) seg_out[row] = off + 1;                    // the reference's ids are 1-based (img_00001.t7)
        if (labels_out && x == 0) labels_out[v] = labels[vid];
    }
    const float *__restrict__ src = store + (size_t)(first_row[vid] + off) * D;
    float *__restrict__ dst = out + (size_t)row * D;
    if ((D & 3) == 0) {
        const float4 *__restrict__ s4 = reinterpret_cast<const float4 *>(src);
        float4 *__restrict__ d4 = reinterpret_cast<float4 *>(dst);
        uint2 *__restrict__ t2 = out16 ? out16 + (size_t)row * (D / 4) : nullptr;   // bf16 twin row (TA3N_FLAG_BF16_STORE)
        for (int i = threadIdx.x; i < D / 4; i += 256) {
            const float4 v4 = s4[i];
            d4[i] = v4;
            if (t2) {
                const unsigned h0 = pack_bf16(v4.x, v4.y), h1 = pack_bf16(v4.z, v4.w);
                t2[i] = make_uint2(h0, h1);
                if (pair_delta) t2[pair_delta / 2 + i] = make_uint2(pack_bf16_lo(v4.x, v4.y, h0), pack_bf16_lo(v4.z, v4.w, h1));   // (uint2 = 2 floats)
            }
        }
    } else {
        for (int i = threadIdx.x; i < D; i += 256) dst[i] = src[i];
    }
}
__global__ __launch_bounds__(256) void gather_segments_kernel(const float *__restrict__ store, const int64_t *__restrict__ first_row,
                                                              const int32_t *__restrict__ num_frames, const int32_t *__restrict__ labels,
                                                              const int32_t *__restrict__ video_ids, int T, int D,
                                                              float *__restrict__ out, int32_t *__restrict__ labels_out,
                                                              int32_t *__restrict__ seg_out, uint2 *__restrict__ out16, int64_t pair_delta) {
    gather_row_f32((int)blockIdx.x, store, first_row, num_frames, labels, video_ids, T, D, out, labels_out, seg_out, out16, pair_delta);
}

// The same batch assembly from a bf16 packed store ("fp16/bf16 memory-mapped blob"): a quarter of the
// bytes of the fp32 store + twin path when the step reads bf16 twins (2 B in, 2 B out per element; the fp32 rows are
// written only if the caller wants them).  One workgroup per output row; D % 8 == 0.
__device__ __forceinline__ void gather_row_bf16(int row, const uint4 *__restrict__ store, const int64_t *__restrict__ first_row,
                                                const int32_t *__restrict__ num_frames, const int32_t *__restrict__ labels,
                                                const int32_t *__restrict__ video_ids, int T, int D,
                                                float4 *__restrict__ out, int32_t *__restrict__ labels_out,
                                                uint4 *__restrict__ out16, int64_t pair_delta) {
    const int v = row / T, x = row - v * T;
    const int vid = video_ids[v];
    const int nf = num_frames[vid];
    int off;
    if (nf >= T) {
        const double tick = (double)nf / (double)T;             // dataset.py:103-116, float64 like the reference's Python
        off = (int)(tick / 2.0 + tick * (double)x);
    } else {
        off = x < nf ? x : nf - 1;
    }
    if (threadIdx.x == 0 && labels_out && x == 0) labels_out[v] = labels[vid];
    const uint4 *__restrict__ src = store + (size_t)(first_row[vid] + off) * (D / 8);
    for (int i = threadIdx.x; i < D / 8; i += 256) {
        const uint4 q = src[i];
        if (out16) {
            out16[(size_t)row * (D / 8) + i] = q;
            if (pair_delta) out16[pair_delta / 4 + (size_t)row * (D / 8) + i] = make_uint4(0u, 0u, 0u, 0u);   // bf16 rows are their own hi plane: lo = 0
        }
        if (out) {
            out[(size_t)row * (D / 4) + 2 * i] = make_float4(__builtin_bit_cast(float, q.x << 16), __builtin_bit_cast(float, q.x & 0xFFFF0000u),
                                                             __builtin_bit_cast(float, q.y << 16), __builtin_bit_cast(float, q.y & 0xFFFF0000u));
            out[(size_t)row * (D / 4) + 2 * i + 1] = make_float4(__builtin_bit_cast(float, q.z << 16), __builtin_bit_cast(float, q.z & 0xFFFF0000u),
                                                                 __builtin_bit_cast(float, q.w << 16), __builtin_bit_cast(float, q.w & 0xFFFF0000u));
        }
    }
}
__global__ __launch_bounds__(256) void gather_segments_bf16_kernel(const uint4 *__restrict__ store, const int64_t *__restrict__ first_row,
                                                                   const int32_t *__restrict__ num_frames, const int32_t *__restrict__ labels,
                                                                   const int32_t *__restrict__ video_ids, int T, int D,
                                                                   float4 *__restrict__ out, int32_t *__restrict__ labels_out,
                                                                   uint4 *__restrict__ out16, int64_t pair_delta) {
    gather_row_bf16((int)blockIdx.x, store, first_row, num_frames, labels, video_ids, T, D, out, labels_out, out16, pair_delta);
}

// Validation metrics of main.validate / test_models.py (reference main.py:707-735, 809-822; test_models.py:155-198)
// over the first n source rows of Y: cross-entropy sum, top-1 / top-5 hits (torch.topk order: ties go to the lower
// class index) and the confusion matrix (rows = label, cols = argmax), ACCUMULATED into ws["metrics"] / ws["confusion"]
// so a validation epoch needs no host synchronisation.  One workgroup, one wave per video, fixed summation order.
__global__ __launch_bounds__(1024) void eval_metrics_kernel(Geom g, float *__restrict__ ws, int n, int reset) {
    __shared__ float part[16][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, C = g.C;
    const int *__restrict__ labels = reinterpret_cast<const int *>(ws + g.o_labels);
    int *__restrict__ conf = reinterpret_cast<int *>(ws + g.o_confusion);
    if (reset)
        for (int i = threadIdx.x; i < C * C; i += 1024) conf[i] = 0;
    __syncthreads();
    float ce = 0.f, t1 = 0.f, t5 = 0.f;
    for (int b = wv; b < n; b += 16) {
        const float y = lane < C ? ws[g.o_Y + (size_t)b * C + lane] : -INFINITY;
        const int lab = labels[b];
        const float m = wave_allreduce_max(y);
        const float ls = logf(wave_allreduce_sum(lane < C ? expf(y - m) : 0.f));
        const float ylab = wave_allreduce_sum(lane == lab ? y : 0.f);
        // rank of the label among the logits, torch.topk order
        const float ahead = wave_allreduce_sum((lane < C && (y > ylab || (y == ylab && lane < lab))) ? 1.f : 0.f);
        // argmax with the same tie rule: the class that has nobody ahead of it
        const float myahead_key = (lane < C && y == m) ? (float)lane : 1e9f;
        float amin = myahead_key;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) amin = fminf(amin, __shfl_xor(amin, off, 64));
        ce += -(ylab - m - ls);
        t1 += ahead < 1.f ? 1.f : 0.f;
        t5 += ahead < 5.f ? 1.f : 0.f;
        if (lane == 0 && lab >= 0 && lab < C) atomicAdd(&conf[lab * C + (int)amin], 1);
    }
    if (lane == 0) { part[wv][0] = ce; part[wv][1] = t1; part[wv][2] = t5; }
    __syncthreads();
    if (threadIdx.x < 3) {
        float v = reset ? 0.f : ws[g.o_metrics + threadIdx.x];
        for (int w = 0; w < 16; ++w) v += part[w][threadIdx.x];
        ws[g.o_metrics + threadIdx.x] = v;
    }
    if (threadIdx.x == 3) ws[g.o_metrics + 3] = (reset ? 0.f : ws[g.o_metrics + 3]) + (float)n;
}

// The same update over floats [4 i0, 4 i1) of the flat prefix with the step's scalars passed by value: lets the host
// split the update so that the first launch of the NEXT step (which only reads the shared frame FC) runs beside the rest
// of this one on a second stream, and keeps it independent of a newer ta3n_set_hyper upload.
__device__ __forceinline__ void sgd_range_body(const Geom &g, float *__restrict__ params, const float *__restrict__ grads,
                                               float *__restrict__ mom, float *__restrict__ ws, int i0, int i1, int norm_off,
                                               int norm_n, float lr, float mu, float wd, float clip, const Hyper &next, int has_next,
                                               int n_blocks, float *red) {
    float4 *__restrict__ p4 = reinterpret_cast<float4 *>(params);
    float4 *__restrict__ m4 = reinterpret_cast<float4 *>(mom);
    const float4 *__restrict__ g4 = reinterpret_cast<const float4 *>(grads);
    const int stride = n_blocks * blockDim.x;
    int i = i0 + blockIdx.x * blockDim.x + threadIdx.x;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f), m = p, gr = p;
    if (i < i1) { p = p4[i]; m = m4[i]; gr = g4[i]; }      // in flight while the norm partials are added up
    float acc = 0.f;
    acc = strided_partial_sum(ws + norm_off, norm_n, (int)threadIdx.x, (int)blockDim.x);
    const float total = sqrtf(block_sum(acc, red));
    float coef = 1.f;
    if (clip > 0.f) coef = fminf(clip / (total + 1e-6f), 1.f);
    // has_next: bit 0 = `next` is valid, bit 1 = this launch records the norm / clip coefficient (one launch per rank and step does: the
    // range that starts at 0, or - sharded update - whichever of the rank's own ranges is launched first)
    if (blockIdx.x == 0 && threadIdx.x == 0 && (i0 == 0 || (has_next & 2))) {
        ws[g.o_grad_norm] = total;
        ws[g.o_grad_norm + 1] = coef;
    }
    // the per-step scalars of the NEXT step ride along (this kernel takes its own by value and never reads ws.hyper)
    if ((has_next & 1) && blockIdx.x == 0 && threadIdx.x < (int)(sizeof(Hyper) / 4))
        reinterpret_cast<uint32_t *>(ws + g.o_hyper)[threadIdx.x] = reinterpret_cast<const uint32_t *>(&next)[threadIdx.x];
    while (i < i1) {
        const int nxt = i + stride;
        float4 pn = p, mn = m, gn = gr;
        if (nxt < i1) { pn = p4[nxt]; mn = m4[nxt]; gn = g4[nxt]; }
        float gg[4] = {gr.x, gr.y, gr.z, gr.w};
        float pp[4] = {p.x, p.y, p.z, p.w};
        float mm[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float d = fmaf(wd, pp[e], gg[e] * coef);
            mm[e] = fmaf(mu, mm[e], d);
            d = fmaf(mu, mm[e], d);
            pp[e] = fmaf(-lr, d, pp[e]);
        }
        p4[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
        m4[i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
        if (g.o_p16 >= 0) {
            const unsigned h0 = pack_bf16(pp[0], pp[1]), h1 = pack_bf16(pp[2], pp[3]);
            reinterpret_cast<uint2 *>(ws + g.o_p16)[i] = make_uint2(h0, h1);
            if (g.pair_delta) reinterpret_cast<uint2 *>(ws + g.o_p16 + g.pair_delta)[i] = make_uint2(pack_bf16_lo(pp[0], pp[1], h0), pack_bf16_lo(pp[2], pp[3], h1));
        }
        i = nxt; p = pn; m = mn; gr = gn;
    }
}
__global__ __launch_bounds__(256) void sgd_range_kernel(Geom g, float *__restrict__ params, const float *__restrict__ grads,
                                                        float *__restrict__ mom, float *__restrict__ ws, int i0, int i1, int norm_off,
                                                        int norm_n, float lr, float mu, float wd, float clip, Hyper next, int has_next) {
    __shared__ float red[8];
    sgd_range_body(g, params, grads, mom, ws, i0, i1, norm_off, norm_n, lr, mu, wd, clip, next, has_next, (int)gridDim.x, red);
}

// The launch that opens a pipelined step WITH its batch assembly (ta3n_train_steps with feeds): workgroups [0, sgd_blocks) are
// sgd_range_kernel's - the previous step's update of the shared frame FC, this step's scalars - and the rest assemble the step's batch, one
// input row each (gather_row_*: the source half's rows, then the target half's).  The two jobs touch disjoint memory (parameters / momentum /
// scalars against input rows / labels) and both must be done before the step's first GEMM launch: as launches of their own the two gathers
// cost the step two boundaries (fresh-batch step +8 us over the resident-batch step at the headline shape).
struct FeedHalf {
    const void *store;
    const int64_t *first_row;
    const int32_t *num_frames, *labels, *video_ids;
    float *out;                  // fp32 input rows of this half (nullptr: a bf16 store feeding a plan that reads twins only)
    int32_t *labels_out;         // nullptr: target half
    float *twin;                 // bf16 twin rows of this half (nullptr: the plan keeps none)
    int32_t rows, bf16;          // ids_per_step * T; the store holds bf16
};
struct FeedPair {
    FeedHalf half[2];
    int64_t pair_delta;
    int32_t T, D;
};
__global__ __launch_bounds__(256) void sgd_open_feed_kernel(Geom g, float *__restrict__ params, const float *__restrict__ grads,
                                                            float *__restrict__ mom, float *__restrict__ ws, int i0, int i1, int norm_off,
                                                            int norm_n, float lr, float mu, float wd, float clip, Hyper next, int has_next,
                                                            int sgd_blocks, FeedPair fp) {
    __shared__ float red[8];
    if ((int)blockIdx.x < sgd_blocks) {
        sgd_range_body(g, params, grads, mom, ws, i0, i1, norm_off, norm_n, lr, mu, wd, clip, next, has_next, sgd_blocks, red);
        return;
    }
    int row = (int)blockIdx.x - sgd_blocks;
    const int h = row < fp.half[0].rows ? 0 : 1;
    if (h) row -= fp.half[0].rows;
    const FeedHalf &f = fp.half[h];
    if (f.bf16)
        gather_row_bf16(row, static_cast<const uint4 *>(f.store), f.first_row, f.num_frames, f.labels, f.video_ids, fp.T, fp.D,
                        reinterpret_cast<float4 *>(f.out), f.labels_out, reinterpret_cast<uint4 *>(f.twin), f.twin ? fp.pair_delta : 0);
    else
        gather_row_f32(row, static_cast<const float *>(f.store), f.first_row, f.num_frames, f.labels, f.video_ids, fp.T, fp.D,
                       f.out, f.labels_out, nullptr, reinterpret_cast<uint2 *>(f.twin), f.twin ? fp.pair_delta : 0);
}

// Closes a fused-update step (gemm_tiles with SgdSide::p_new: every gradient tile already applied the Nesterov step to its own
// parameters with the clip coefficient taken as 1).  Adds up the step's norm partials in the fixed order of sgd_kernel; leaves the
// norm / coefficient and - like sgd_range_kernel - the NEXT step's scalars in the workspace; and if the gradient norm did exceed
// clip_gradient (coef < 1), corrects parameters and momentum: the update is linear in the gradient,
//   p(coef) = p(1) + lr (1 + mu) (1 - coef) g,   m(coef) = m(1) - (1 - coef) g
// (clip_grad_norm_ + SGD, main.py:578-583; equal to the direct computation up to fp32 rounding of these two operations).
__global__ __launch_bounds__(256) void sgd_fixup_kernel(Geom g, float *__restrict__ params, const float *__restrict__ grads,
                                                        float *__restrict__ mom, float *__restrict__ ws, float *__restrict__ p16, int n4,
                                                        float lr, float mu, float clip, Hyper next, int has_next) {
    __shared__ float red[8];
    float acc = 0.f;
    acc = strided_partial_sum(ws + g.o_sumsq, g.n_sumsq, (int)threadIdx.x, (int)blockDim.x);
    const float total = sqrtf(block_sum(acc, red));
    float coef = 1.f;
    if (clip > 0.f) coef = fminf(clip / (total + 1e-6f), 1.f);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ws[g.o_grad_norm] = total;
        ws[g.o_grad_norm + 1] = coef;
    }
    if (has_next && blockIdx.x == 0 && threadIdx.x < (int)(sizeof(Hyper) / 4))
        reinterpret_cast<uint32_t *>(ws + g.o_hyper)[threadIdx.x] = reinterpret_cast<const uint32_t *>(&next)[threadIdx.x];
    if (coef >= 1.f) return;                       // (uniform over the grid) the speculation held: nothing to correct
    const float c = 1.f - coef, a = lr * (1.f + mu) * c;
    float4 *__restrict__ p4 = reinterpret_cast<float4 *>(params);
    float4 *__restrict__ m4 = reinterpret_cast<float4 *>(mom);
    const float4 *__restrict__ g4 = reinterpret_cast<const float4 *>(grads);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
        const float4 gr = g4[i];
        float4 p = p4[i], m = m4[i];
        p.x = fmaf(a, gr.x, p.x); p.y = fmaf(a, gr.y, p.y); p.z = fmaf(a, gr.z, p.z); p.w = fmaf(a, gr.w, p.w);
        m.x = fmaf(-c, gr.x, m.x); m.y = fmaf(-c, gr.y, m.y); m.z = fmaf(-c, gr.z, m.z); m.w = fmaf(-c, gr.w, m.w);
        p4[i] = p; m4[i] = m;
        if (p16 != nullptr) reinterpret_cast<uint2 *>(p16)[i] = make_uint2(pack_bf16(p.x, p.y), pack_bf16(p.z, p.w));
    }
}

__global__ void to_bf16_kernel(const float4 *__restrict__ src, uint2 *__restrict__ dst, int64_t n4) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = src[i];
        dst[i] = make_uint2(pack_bf16(v.x, v.y), pack_bf16(v.z, v.w));
    }
}

__global__ void to_bf16_pair_kernel(const float4 *__restrict__ src, uint2 *__restrict__ hi, uint2 *__restrict__ lo, int64_t n4) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = src[i];
        const unsigned h0 = pack_bf16(v.x, v.y), h1 = pack_bf16(v.z, v.w);
        hi[i] = make_uint2(h0, h1);
        lo[i] = make_uint2(pack_bf16_lo(v.x, v.y, h0), pack_bf16_lo(v.z, v.w, h1));
    }
}

// The per-step scalars travel as a kernel argument (copied at launch time): no staging buffer whose reuse would have to be
// fenced against the copy, however far the host runs ahead of the stream.
__global__ void set_hyper_kernel(Hyper *__restrict__ dst, Hyper h) {
    if (threadIdx.x == 0) *dst = h;
}

__global__ void fill_kernel(float *__restrict__ dst, float v, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = v;
}

}  // namespace

namespace ta3n {

#define TA3N_DISPATCH_Q(KERNEL, grid, stream, ...)                                                    \
    switch (g.NB / 64) {                                                                              \
        case 1: hipLaunchKernelGGL((KERNEL<1>), grid, dim3(256), 0, stream, __VA_ARGS__); break;      \
        case 2: hipLaunchKernelGGL((KERNEL<2>), grid, dim3(256), 0, stream, __VA_ARGS__); break;      \
        case 4: hipLaunchKernelGGL((KERNEL<4>), grid, dim3(256), 0, stream, __VA_ARGS__); break;      \
        case 8: hipLaunchKernelGGL((KERNEL<8>), grid, dim3(256), 0, stream, __VA_ARGS__); break;      \
        default: return -1;                                                                           \
    }

int launch_pool_fwd(const Geom &g, const Ptrs &ptrs, hipStream_t stream) {
    const dim3 grid((g.B + 3) / 4);
    TA3N_DISPATCH_Q(pool_fwd_kernel, grid, stream, g, ptrs)
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_pool_bwd(const Geom &g, const Ptrs &ptrs, hipStream_t stream) {
    const dim3 grid((g.B + 3) / 4);
    TA3N_DISPATCH_Q(pool_bwd_kernel, grid, stream, g, ptrs)
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_loss(const Geom &g, const Ptrs &ptrs, hipStream_t stream) {
    const int rows = g.B * (1 + g.n_rel + g.T);
    hipLaunchKernelGGL(loss_kernel, dim3((rows + 255) / 256), dim3(256), 0, stream, g, ptrs);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_pool_avg_fwd(const Geom &g, const Ptrs &ptrs, hipStream_t stream) {
    hipLaunchKernelGGL(pool_avg_fwd_kernel, dim3(g.B), dim3(256), 0, stream, g, ptrs);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---- use_bn AdaBN / AutoDIAL: BatchNorm1d per domain between the shared frame FC and its ReLU (models.py:490-543, 569-570) ----
// One workgroup = BN_COLS (= 4) feature columns of one domain (blockIdx.y: 0 source rows [0, Bs T), 1 target rows [Bs T, B T)), 1 024
// threads; thread t owns ROWS t, t + 1024, ... and moves the four columns of a row as ONE 16-byte access (row-contiguous float4 loads and
// stores, one 8-byte twin store) - the 8 x 32 layout of the first fused version read 32-byte row pieces four bytes per lane and ran on half
// the CUs with one wave per SIMD (11.5 + 9.8 us for a 2 MB matrix).  Column sums: DPP butterfly inside a wave, then the sixteen waves'
// partials in wave order from LDS (fixed order: bitwise reproducible).  Statistics as torch's CPU kernel takes them: the mean first, then the
// sum of squared deviations (two passes); eps = 1e-5 (nn.BatchNorm1d default).
constexpr float BN_EPS = 1e-5f;
constexpr int BN_THREADS = 1024;
constexpr int BN_WAVES = BN_THREADS / 64;
constexpr int BN_KEEP = 5;      // rows per thread that stay in registers between the passes: batches of up to 5 120 frame rows per domain (configs[3]: 4 608)
static_assert(BN_COLS == 4, "the BatchNorm launches move one float4 per row");

struct BnRow { float v[BN_COLS]; };

// the four columns [c0, c0 + 4) of one row: one 16-byte access (ta3n_plan_create refuses use_bn with fc_dim % 4 != 0; regions start 256-byte aligned)
__device__ __forceinline__ BnRow bn_load(const float *__restrict__ row, int c0) {
    BnRow o;
    const float4 t = *reinterpret_cast<const float4 *>(row + c0);
    o.v[0] = t.x; o.v[1] = t.y; o.v[2] = t.z; o.v[3] = t.w;
    return o;
}
__device__ __forceinline__ void bn_store(float *__restrict__ row, unsigned short *__restrict__ tw, int pair_delta, int c0, const BnRow &y) {
    *reinterpret_cast<float4 *>(row + c0) = make_float4(y.v[0], y.v[1], y.v[2], y.v[3]);
    if (tw) {      // bf16 twin (TA3N_FLAG_BF16_STORE): the next GEMM launch reads it as an operand (ta3n_plan.cpp: add_bf16_twins)
        const unsigned h0 = pack_bf16(y.v[0], y.v[1]), h1 = pack_bf16(y.v[2], y.v[3]);
        *reinterpret_cast<uint2 *>(tw + c0) = make_uint2(h0, h1);
        if (pair_delta) *reinterpret_cast<uint2 *>(tw + c0 + 2 * (size_t)pair_delta) = make_uint2(pack_bf16_lo(y.v[0], y.v[1], h0), pack_bf16_lo(y.v[2], y.v[3], h1));
    }
}
// NV per-thread values summed over the workgroup; every thread receives every total (wave butterfly, then waves 0 .. 15 in order)
template <int NV>
__device__ __forceinline__ void bn_block_sum(float (&v)[NV], float *red) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = wave_allreduce_sum(v[k]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) red[wave * NV + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < BN_WAVES; ++w) t += red[w * NV + k];
        v[k] = t;
    }
}

// Which column group a workgroup owns.  Workgroup b of a launch runs on XCD b mod 8 and every XCD has an L2 of its own: with the groups dealt
// in launch order the eight 16-byte pieces of one 128-byte line of a row belonged to eight XCDs - every line fetched eight times over the
// fabric and written back as eight masked pieces.  Dealt this way the eight groups of a line share an XCD (its L2 fetches the line once and
// merges the stores).  Needs the group count to be a multiple of 64; otherwise launch order.
__device__ __forceinline__ int bn_column_group(int b, int n_groups) {
    if (n_groups & 63) return b;
    const int x = b & 7, k = b >> 3;
    return ((x + 8 * (k >> 3)) << 3) + (k & 7);
}

__global__ __launch_bounds__(BN_THREADS) void bn_shared_fwd_kernel(Geom g, Ptrs ptrs) {
    __shared__ float red[BN_WAVES * BN_COLS];
    const Hyper *__restrict__ hy = reinterpret_cast<const Hyper *>(ptrs.ws + g.o_hyper);
    float *__restrict__ ws = ptrs.ws;
    const int dom = blockIdx.y, tid = threadIdx.x, cg = bn_column_group((int)blockIdx.x, (int)gridDim.x), c0 = cg * BN_COLS;
    const int row0 = dom == 0 ? 0 : g.Bs * g.T, n = dom == 0 ? g.Bs * g.T : g.Bt * g.T, F = g.F;
    if (n == 0) return;
    const float *__restrict__ z = ws + g.o_Z0 + (size_t)row0 * F;
    // Everything else the launch reads - the step's scalars, the affine pair, the running statistics - is requested HERE, beside the column
    // slab: `hy` and the statistics live in the workspace this kernel stores to, so a read placed behind a store is a new round trip behind
    // that store's acknowledgement (vmcnt counts both on this target); the first version paid three of those one after the other.
    const int train = hy->train;
    const float p_drop = hy->p_drop_i;
    const uint32_t seed = hy->seed_i;
    float w[BN_COLS], b[BN_COLS], run_m[BN_COLS], run_v[BN_COLS];
    {
        const float *run = ws + g.o_bn_run + (size_t)dom * 2 * F;
#pragma unroll
        for (int e = 0; e < BN_COLS; ++e) {
            w[e] = ptrs.p[g.p_bn_w[dom] + c0 + e];
            b[e] = ptrs.p[g.p_bn_b[dom] + c0 + e];
            run_m[e] = run[c0 + e];
            run_v[e] = run[F + c0 + e];
        }
    }
    // Up to BN_KEEP rows per thread stay in registers between the three passes - mean, variance, apply - so the column slab crosses the
    // memory system once; taller batches stream (same additions in the same order either way).
    BnRow zreg[BN_KEEP];
    const bool keep = n <= BN_THREADS * BN_KEEP;      // (uniform)
    if (keep) {
#pragma unroll
        for (int j = 0; j < BN_KEEP; ++j) {
            const int i = tid + j * BN_THREADS;
            if (i < n) zreg[j] = bn_load(z + (size_t)i * F, c0);
            else {
#pragma unroll
                for (int e = 0; e < BN_COLS; ++e) zreg[j].v[e] = 0.f;
            }
        }
    }
    float mean[BN_COLS], invstd[BN_COLS];
    if (train) {
        float s[BN_COLS] = {0.f, 0.f, 0.f, 0.f};
        if (keep) {
#pragma unroll
            for (int j = 0; j < BN_KEEP; ++j) if (tid + j * BN_THREADS < n) {
#pragma unroll
                for (int e = 0; e < BN_COLS; ++e) s[e] += zreg[j].v[e];
            }
        } else for (int i = tid; i < n; i += BN_THREADS) {
            const BnRow t = bn_load(z + (size_t)i * F, c0);
#pragma unroll
            for (int e = 0; e < BN_COLS; ++e) s[e] += t.v[e];
        }
        bn_block_sum<BN_COLS>(s, red);
#pragma unroll
        for (int e = 0; e < BN_COLS; ++e) mean[e] = s[e] / (float)n;
        float q[BN_COLS] = {0.f, 0.f, 0.f, 0.f};
        if (keep) {
#pragma unroll
            for (int j = 0; j < BN_KEEP; ++j) if (tid + j * BN_THREADS < n) {
#pragma unroll
                for (int e = 0; e < BN_COLS; ++e) { const float d = zreg[j].v[e] - mean[e]; q[e] = fmaf(d, d, q[e]); }
            }
        } else for (int i = tid; i < n; i += BN_THREADS) {
            const BnRow t = bn_load(z + (size_t)i * F, c0);
#pragma unroll
            for (int e = 0; e < BN_COLS; ++e) { const float d = t.v[e] - mean[e]; q[e] = fmaf(d, d, q[e]); }
        }
        bn_block_sum<BN_COLS>(q, red);
        float *st = ws + g.o_bn_batch + (size_t)dom * 3 * F;
        // nn.BatchNorm1d's buffer update, on the device (round 6: a K-step call has no host between its steps): momentum 0.1, unbiased
        // batch variance (models.py:195-198 modules in train mode); the eval-mode branch below reads the same region
        float *run = ws + g.o_bn_run + (size_t)dom * 2 * F;
        const float unb = (float)n / (float)(n > 1 ? n - 1 : 1);
#pragma unroll
        for (int e = 0; e < BN_COLS; ++e) {
            const float var = q[e] / (float)n;
            invstd[e] = 1.f / sqrtf(var + BN_EPS);
            if (tid == 0) {
                const int c = c0 + e;
                st[c] = mean[e]; st[F + c] = var; st[2 * F + c] = invstd[e];
                run[c] = run_m[e] * 0.9f + 0.1f * mean[e];
                run[F + c] = run_v[e] * 0.9f + 0.1f * (var * unb);
            }
        }
    } else {
#pragma unroll
        for (int e = 0; e < BN_COLS; ++e) { mean[e] = run_m[e]; invstd[e] = 1.f / sqrtf(run_v[e] + BN_EPS); }
    }
    const bool drop = train && p_drop > 0.f;
    const float inv_keep = drop ? (p_drop < 1.f ? 1.f / (1.f - p_drop) : 0.f) : 1.f;      // (hyper_scale(hy, SK_INV_KEEP_I) on the scalars read above)
    float *__restrict__ out = ws + g.o_F1 + (size_t)row0 * F;
    unsigned short *__restrict__ tw = g.o_ws16 >= 0 ? reinterpret_cast<unsigned short *>(ws + g.o_ws16) + g.o_F1 + (size_t)row0 * F : nullptr;
    auto apply = [&](int i, const BnRow &zv) {
        BnRow y;
#pragma unroll
        for (int e = 0; e < BN_COLS; ++e) {
            float t = fmaf((zv.v[e] - mean[e]) * invstd[e], w[e], b[e]);
            t = fmaxf(t, 0.f);
            if (drop) t *= keep_mask(seed, (uint32_t)((row0 + i) * F + c0 + e), p_drop);
            y.v[e] = t * inv_keep;
        }
        bn_store(out + (size_t)i * F, tw ? tw + (size_t)i * F : nullptr, g.pair_delta, c0, y);
    };
    if (keep) {
#pragma unroll
        for (int j = 0; j < BN_KEEP; ++j) { const int i = tid + j * BN_THREADS; if (i < n) apply(i, zreg[j]); }
    } else {
        for (int i = tid; i < n; i += BN_THREADS) apply(i, bn_load(z + (size_t)i * F, c0));
    }
}

// gZ1 = dL/d(BatchNorm output) (ReLU mask and dropout already applied by the launch that made it)  ->
// d weight = sum g xhat, d bias = sum g, gZ0 = weight invstd (g - mean(g) - xhat mean(g xhat))   (train mode)
__global__ __launch_bounds__(BN_THREADS) void bn_shared_bwd_kernel(Geom g, Ptrs ptrs) {
    __shared__ float red[BN_WAVES * 2 * BN_COLS];
    float *__restrict__ ws = ptrs.ws;
    const int dom = blockIdx.y, tid = threadIdx.x, cg = bn_column_group((int)blockIdx.x, (int)gridDim.x), c0 = cg * BN_COLS;
    const int row0 = dom == 0 ? 0 : g.Bs * g.T, n = dom == 0 ? g.Bs * g.T : g.Bt * g.T, F = g.F;
    // fused step: this workgroup's share of the gradient norm (sum of squares of the 2 x BN_COLS BatchNorm gradients it writes) goes to ITS
    // slot at the end of ws["sumsq"] (ta3n_plan.cpp: the last 2 * gridDim.x slots) - the fused optimiser adds the slots in a fixed order
    float *slot = g.n_sumsq > 0 ? ws + g.o_sumsq + g.n_sumsq - 2 * (int)gridDim.x + dom * (int)gridDim.x + cg : nullptr;
    if (n == 0) {
        if (slot && tid == 0) *slot = 0.f;
        return;
    }
    const float *__restrict__ z = ws + g.o_Z0 + (size_t)row0 * F;
    const float *__restrict__ gy = ws + g.o_gZ1 + (size_t)row0 * F;
    const float *st = ws + g.o_bn_batch + (size_t)dom * 3 * F;
    float mean[BN_COLS], invstd[BN_COLS], wgt[BN_COLS];      // (all requested beside the slabs: a read behind this kernel's stores would wait for them)
#pragma unroll
    for (int e = 0; e < BN_COLS; ++e) {
        mean[e] = st[c0 + e];
        invstd[e] = st[2 * F + c0 + e];
        wgt[e] = ptrs.p[g.p_bn_w[dom] + c0 + e];
    }
    BnRow greg[BN_KEEP], xreg[BN_KEEP];      // (as in the forward launch: gradient and normalised input of up to BN_KEEP rows per thread stay in registers)
    const bool keep = n <= BN_THREADS * BN_KEEP;
    float sums[2 * BN_COLS] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // [e]: sum g, [BN_COLS + e]: sum g xhat
    if (keep) {
#pragma unroll
        for (int j = 0; j < BN_KEEP; ++j) {
            const int i = tid + j * BN_THREADS;
            if (i < n) {
                greg[j] = bn_load(gy + (size_t)i * F, c0);
                xreg[j] = bn_load(z + (size_t)i * F, c0);
            } else {
#pragma unroll
                for (int e = 0; e < BN_COLS; ++e) { greg[j].v[e] = 0.f; xreg[j].v[e] = 0.f; }
            }
        }
#pragma unroll
        for (int j = 0; j < BN_KEEP; ++j) if (tid + j * BN_THREADS < n) {
#pragma unroll
            for (int e = 0; e < BN_COLS; ++e) {
                xreg[j].v[e] = (xreg[j].v[e] - mean[e]) * invstd[e];
                sums[e] += greg[j].v[e];
                sums[BN_COLS + e] = fmaf(greg[j].v[e], xreg[j].v[e], sums[BN_COLS + e]);
            }
        }
    } else for (int i = tid; i < n; i += BN_THREADS) {
        const BnRow gv = bn_load(gy + (size_t)i * F, c0), zv = bn_load(z + (size_t)i * F, c0);
#pragma unroll
        for (int e = 0; e < BN_COLS; ++e) {
            sums[e] += gv.v[e];
            sums[BN_COLS + e] = fmaf(gv.v[e], (zv.v[e] - mean[e]) * invstd[e], sums[BN_COLS + e]);
        }
    }
    bn_block_sum<2 * BN_COLS>(sums, red);
    if (tid == 0) {
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < BN_COLS; ++e) {      // (added in column order)
            q += fmaf(sums[BN_COLS + e], sums[BN_COLS + e], sums[e] * sums[e]);
            ptrs.g[g.p_bn_w[dom] + c0 + e] = sums[BN_COLS + e];
            ptrs.g[g.p_bn_b[dom] + c0 + e] = sums[e];
        }
        if (slot) *slot = q;
    }
    float k[BN_COLS], mg[BN_COLS], mgx[BN_COLS];
#pragma unroll
    for (int e = 0; e < BN_COLS; ++e) {
        k[e] = wgt[e] * invstd[e]; mg[e] = sums[e] / (float)n; mgx[e] = sums[BN_COLS + e] / (float)n;
    }
    float *__restrict__ out = ws + g.o_gZ0 + (size_t)row0 * F;
    unsigned short *__restrict__ tw = g.o_ws16 >= 0 ? reinterpret_cast<unsigned short *>(ws + g.o_ws16) + g.o_gZ0 + (size_t)row0 * F : nullptr;
    auto emit = [&](int i, const BnRow &gv, const BnRow &xh) {      // (the twin of gZ0: the shared-FC weight-gradient launch reads it)
        BnRow v;
#pragma unroll
        for (int e = 0; e < BN_COLS; ++e) v.v[e] = k[e] * (gv.v[e] - mg[e] - xh.v[e] * mgx[e]);
        bn_store(out + (size_t)i * F, tw ? tw + (size_t)i * F : nullptr, g.pair_delta, c0, v);
    };
    if (keep) {
#pragma unroll
        for (int j = 0; j < BN_KEEP; ++j) { const int i = tid + j * BN_THREADS; if (i < n) emit(i, greg[j], xreg[j]); }
    } else {
        for (int i = tid; i < n; i += BN_THREADS) {
            BnRow xh = bn_load(z + (size_t)i * F, c0);
#pragma unroll
            for (int e = 0; e < BN_COLS; ++e) xh.v[e] = (xh.v[e] - mean[e]) * invstd[e];
            emit(i, bn_load(gy + (size_t)i * F, c0), xh);
        }
    }
}

int launch_bn_shared_fwd(const Geom &g, const Ptrs &ptrs, hipStream_t stream) {
    hipLaunchKernelGGL(bn_shared_fwd_kernel, dim3((g.F + BN_COLS - 1) / BN_COLS, 2), dim3(BN_THREADS), 0, stream, g, ptrs);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
int launch_bn_shared_bwd(const Geom &g, const Ptrs &ptrs, hipStream_t stream) {
    hipLaunchKernelGGL(bn_shared_bwd_kernel, dim3((g.F + BN_COLS - 1) / BN_COLS, 2), dim3(BN_THREADS), 0, stream, g, ptrs);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---- ens_DA MCD: what main.py does between its launches, without a framework in between (main.py:447-448, 548-562; loss.py:29-30) ----
// A row of <= 64 class logits on one wave (lane c = class c): softmax, log_softmax and entropy the way torch takes them.
struct RowSoft { float p, lp, H; };
__device__ __forceinline__ RowSoft row_soft(float z, bool on) {
    const float m = wave_allreduce_max(on ? z : -INFINITY);
    const float e = on ? expf(z - m) : 0.f;
    const float sum = wave_allreduce_sum(e);
    RowSoft r;
    r.p = e / sum;
    r.lp = on ? z - m - logf(sum) : 0.f;
    r.H = -wave_allreduce_sum(on ? r.p * r.lp : 0.f);
    return r;
}
// + CrossEntropy(out_source_2, label) over the valid source rows (main.py:447-448): its logit gradient to gY2 (rows without a label: 0), the
// per-row loss terms to part[b]; with the attentive entropy on, the target rows of gY are cleared - that term moves to the second pass
// (mcd_second_loss_kernel).  One wave per video.
__global__ __launch_bounds__(256) void mcd_source_loss_kernel(Geom g, float *__restrict__ ws, float *__restrict__ part) {
    const Hyper *__restrict__ hy = reinterpret_cast<const Hyper *>(ws + g.o_hyper);
    const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= g.B) return;
    const int C = g.C, ns = hy->valid_source;
    const float inv = hy->inv_n_cls;
    const bool on = lane < C;
    float g2 = 0.f, term = 0.f;
    if (b < ns) {
        const int label = reinterpret_cast<const int *>(ws + g.o_labels)[b];
        const RowSoft r = row_soft(on ? ws[g.o_Y2 + (size_t)b * C + lane] : 0.f, on);
        g2 = (expf(r.lp) - (lane == label ? 1.f : 0.f)) * inv;
        term = -wave_allreduce_sum(lane == label ? r.lp : 0.f);
    }
    if (on) {
        ws[g.o_gY2 + (size_t)b * C + lane] = g2;
        if ((g.flags & TA3N_FLAG_ATTN_ENTROPY) && b >= g.Bs) ws[g.o_gY + (size_t)b * C + lane] = 0.f;
    }
    if (lane == 0) part[b] = term;
}
// The second, reversed pass's loss (main.py:548-562): loss_s = -mean |softmax(out_target) - softmax(out_target_2)| over the valid target
// rows of the GLOBAL batch (loss.py:29-30) and - the reference REBINDS out_target before it assembles the attentive entropy - the target
// half of that loss on THIS pass's logits, weighted by the FIRST pass's video-domain logits: gY / gY2 of the second workspace, the first
// pass's gPv moved by d(e_new - e_old) / d Pv, per-row terms {sum |dp|, w H(y second), w H(y first)} to part[r][0..2].  One wave per row.
__global__ __launch_bounds__(256) void mcd_second_loss_kernel(Geom g, float *__restrict__ ws, float *__restrict__ ws2, float inv_count,
                                                              float *__restrict__ part) {
    const Hyper *__restrict__ hy = reinterpret_cast<const Hyper *>(ws + g.o_hyper);
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nt = hy->valid_target, C = g.C;
    if (r >= nt) return;
    const int b = g.Bs + r;
    const bool on = lane < C;
    const RowSoft a = row_soft(on ? ws2[g.o_Y + (size_t)b * C + lane] : 0.f, on);
    const RowSoft c = row_soft(on ? ws2[g.o_Y2 + (size_t)b * C + lane] : 0.f, on);
    const float dp = a.p - c.p;
    const float sgn = dp > 0.f ? 1.f : dp < 0.f ? -1.f : 0.f;
    // loss = -inv_count * sum |dp|:  d / d p = -inv_count sgn,  d / d p2 = +inv_count sgn;  softmax backward  g_k = p_k (d_k - sum_c p_c d_c)
    const float d1 = on ? -inv_count * sgn : 0.f;
    float g1 = a.p * (d1 - wave_allreduce_sum(a.p * d1));
    const float g2 = c.p * (-d1 - wave_allreduce_sum(c.p * -d1));
    const float sabs = wave_allreduce_sum(on ? fabsf(dp) : 0.f);
    float e_new = 0.f, e_old = 0.f;
    if (g.flags & TA3N_FLAG_ATTN_ENTROPY) {
        const float scale = hy->gamma * hy->inv_n_ent;
        const RowSoft f = row_soft(on ? ws[g.o_Y + (size_t)b * C + lane] : 0.f, on);
        const Soft2 pv = soft2(ws[g.o_Pv + (size_t)b * 2], ws[g.o_Pv + (size_t)b * 2 + 1]);
        const float w = 1.f + pv.H;
        e_new = w * a.H; e_old = w * f.H;
        g1 += on ? scale * w * (-a.p * (a.lp + a.H)) : 0.f;                  // d H(z) / d z_k = -p_k (log p_k + H)
        if (lane < 2) {
            const float pj = lane == 0 ? pv.p0 : pv.p1, lpj = lane == 0 ? pv.lp0 : pv.lp1;
            ws[g.o_gPv + (size_t)b * 2 + lane] += scale * (a.H - f.H) * (-pj * (lpj + pv.H));
        }
    }
    if (on) {
        ws2[g.o_gY + (size_t)b * C + lane] = g1;
        ws2[g.o_gY2 + (size_t)b * C + lane] = g2;
    }
    if (lane == 0) { part[(size_t)r * 3] = sabs; part[(size_t)r * 3 + 1] = e_new; part[(size_t)r * 3 + 2] = e_old; }
}
// out[slot + k] = scale[k] * sum_r part[r * stride + k]  (k < n_out; rows in order: one wave, fixed order)
__global__ __launch_bounds__(64) void mcd_finish_kernel(const float *__restrict__ part, const Geom g, const float *__restrict__ ws, int which,
                                                        float inv_count, float *__restrict__ out) {
    const Hyper *__restrict__ hy = reinterpret_cast<const Hyper *>(ws + g.o_hyper);
    if (which == 0) {      // loss_c2
        const int n = hy->valid_source;
        float acc = 0.f;
        for (int i = threadIdx.x; i < n; i += 64) acc += part[i];
        acc = wave_allreduce_sum(acc);
        if (threadIdx.x == 0) out[0] = acc * hy->inv_n_cls;
        return;
    }
    const int n = hy->valid_target;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int i = threadIdx.x; i < n; i += 64) { s0 += part[(size_t)i * 3]; s1 += part[(size_t)i * 3 + 1]; s2 += part[(size_t)i * 3 + 2]; }
    s0 = wave_allreduce_sum(s0); s1 = wave_allreduce_sum(s1); s2 = wave_allreduce_sum(s2);
    if (threadIdx.x == 0) {
        const float scale = hy->gamma * hy->inv_n_ent;
        const float de = (g.flags & TA3N_FLAG_ATTN_ENTROPY) ? scale * s1 - scale * s2 : 0.f;
        out[1] = -inv_count * s0;                                    // loss_s
        out[2] = de;                                                 // what moving the target rows' entropy term to this pass adds to the total loss
        out[3] = hy->gamma != 0.f ? de / hy->gamma : 0.f;            // ... and to main.py's loss_e
    }
}

int launch_mcd_source_loss(const Geom &g, float *ws, float *part, float *out, hipStream_t stream) {
    hipLaunchKernelGGL(mcd_source_loss_kernel, dim3((g.B + 3) / 4), dim3(256), 0, stream, g, ws, part);
    hipLaunchKernelGGL(mcd_finish_kernel, dim3(1), dim3(64), 0, stream, part, g, ws, 0, 0.f, out);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
int launch_mcd_second_loss(const Geom &g, float *ws, float *ws2, float inv_count, float *part, float *out, hipStream_t stream) {
    if (g.Bt > 0) hipLaunchKernelGGL(mcd_second_loss_kernel, dim3((g.Bt + 3) / 4), dim3(256), 0, stream, g, ws, ws2, inv_count, part);
    hipLaunchKernelGGL(mcd_finish_kernel, dim3(1), dim3(64), 0, stream, part, g, ws, 1, inv_count, out);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_pool_avg_bwd(const Geom &g, const Ptrs &ptrs, hipStream_t stream) {
    hipLaunchKernelGGL(pool_avg_bwd_kernel, dim3(g.B * g.T), dim3(256), 0, stream, g, ptrs, g.o_gHf < 0 ? 1 : 0);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_pool_cls(const Geom &g, const Ptrs &ptrs, hipStream_t stream) {
    hipLaunchKernelGGL(pool_cls_kernel, dim3(g.B), dim3(256), (2 * g.F + 64) * sizeof(float), stream, g, ptrs);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_grad_norm(const Geom &g, const float *grads, float *ws, hipStream_t stream) {
    hipLaunchKernelGGL(grad_norm_kernel, dim3(g.n_norm_blocks), dim3(256), 0, stream, grads, ws + g.o_norm_part,
                       g.live_floats / 4);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// Sharded update (ta3n_sharded_update): sum of squares of the REDUCED gradients this rank owns - two index ranges [a0, a1) and
// [b0, b1) in float4 units - per block, then (shard_norm_finish_kernel, one block) added up in a fixed order into slot `rank` of the
// norm_part region with every other slot zeroed: an all-gather of one float per rank into slots [0, world) completes it, and
// sgd_range_kernel adds the region up the way it always does.
__global__ __launch_bounds__(256) void shard_sumsq_kernel(const float *__restrict__ grads, float *__restrict__ part, int a0, int a1, int b0, int b1) {
    __shared__ float red[8];
    float acc = 0.f;
    const float4 *__restrict__ g4 = reinterpret_cast<const float4 *>(grads);
    const int stride = gridDim.x * blockDim.x, t = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = a0 + t; i < a1; i += stride) {
        const float4 v = g4[i];
        acc = fmaf(v.x, v.x, acc); acc = fmaf(v.y, v.y, acc); acc = fmaf(v.z, v.z, acc); acc = fmaf(v.w, v.w, acc);
    }
    for (int i = b0 + t; i < b1; i += stride) {
        const float4 v = g4[i];
        acc = fmaf(v.x, v.x, acc); acc = fmaf(v.y, v.y, acc); acc = fmaf(v.z, v.z, acc); acc = fmaf(v.w, v.w, acc);
    }
    const float s = block_sum(acc, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void shard_norm_finish_kernel(float *__restrict__ part, int n, int rank) {
    __shared__ float red[8];
    float acc = 0.f;
    for (int k = threadIdx.x; k < n; k += blockDim.x) acc += part[k];
    const float s = block_sum(acc, red);      // (broadcast to every thread; also a barrier: all reads above are done)
    for (int k = threadIdx.x; k < n; k += blockDim.x) part[k] = (k == rank) ? s : 0.f;
}

int launch_shard_sumsq(const Geom &g, const float *grads, float *ws, int64_t a0, int64_t a1, int64_t b0, int64_t b1, int rank, hipStream_t stream) {
    hipLaunchKernelGGL(shard_sumsq_kernel, dim3(g.n_norm_blocks), dim3(256), 0, stream, grads, ws + g.o_norm_part, (int)(a0 / 4), (int)(a1 / 4),
                       (int)(b0 / 4), (int)(b1 / 4));
    hipLaunchKernelGGL(shard_norm_finish_kernel, dim3(1), dim3(256), 0, stream, ws + g.o_norm_part, g.n_norm_blocks, rank);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_sgd(const Geom &g, float *params, const float *grads, float *momentum, float *ws, hipStream_t stream, bool fused_norm) {
    const int n4 = g.live_floats / 4;
    int blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(sgd_kernel, dim3(blocks), dim3(256), 0, stream, g, params, grads, momentum, ws, n4,
                       fused_norm ? g.o_sumsq : g.o_norm_part, fused_norm ? g.n_sumsq : g.n_norm_blocks);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_gather_segments(const float *store, const int64_t *first_row, const int32_t *num_frames, const int32_t *labels,
                           const int32_t *video_ids, int n_videos, int T, int D, float *out, int32_t *labels_out, int32_t *seg_out,
                           float *out_twin, hipStream_t stream, int64_t pair_delta) {
    if (n_videos <= 0) return 0;
    hipLaunchKernelGGL(gather_segments_kernel, dim3(n_videos * T), dim3(256), 0, stream, store, first_row, num_frames, labels, video_ids,
                       T, D, out, labels_out, seg_out, reinterpret_cast<uint2 *>(out_twin), out_twin ? pair_delta : 0);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_gather_segments_bf16(const void *store16, const int64_t *first_row, const int32_t *num_frames, const int32_t *labels,
                                const int32_t *video_ids, int n_videos, int T, int D, float *out, int32_t *labels_out, float *out_twin,
                                hipStream_t stream, int64_t pair_delta) {
    if (n_videos <= 0) return 0;
    hipLaunchKernelGGL(gather_segments_bf16_kernel, dim3(n_videos * T), dim3(256), 0, stream, static_cast<const uint4 *>(store16), first_row,
                       num_frames, labels, video_ids, T, D, reinterpret_cast<float4 *>(out), labels_out, reinterpret_cast<uint4 *>(out_twin),
                       out_twin ? pair_delta : 0);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_eval_metrics(const Geom &g, float *ws, int n, int reset, hipStream_t stream) {
    hipLaunchKernelGGL(eval_metrics_kernel, dim3(1), dim3(1024), 0, stream, g, ws, n, reset);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_sgd_range(const Geom &g, float *params, const float *grads, float *momentum, float *ws, int64_t begin, int64_t end,
                     bool fused_norm, float lr, float mu, float wd, float clip, const Hyper *next, hipStream_t stream, bool write_norm) {
    const int i0 = (int)(begin / 4), i1 = (int)(end / 4);
    if (i1 <= i0) return 0;
    int blocks = (i1 - i0 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    Hyper nh;
    std::memset(&nh, 0, sizeof(nh));
    if (next) nh = *next;
    hipLaunchKernelGGL(sgd_range_kernel, dim3(blocks), dim3(256), 0, stream, g, params, grads, momentum, ws, i0, i1,
                       fused_norm ? g.o_sumsq : g.o_norm_part, fused_norm ? g.n_sumsq : g.n_norm_blocks, lr, mu, wd, clip, nh,
                       (next ? 1 : 0) | (write_norm ? 2 : 0));
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_sgd_open_feed(const Geom &g, float *params, const float *grads, float *momentum, float *ws, int64_t begin, int64_t end,
                         bool fused_norm, float lr, float mu, float wd, float clip, const Hyper *next, const FeedJob feeds[2], hipStream_t stream) {
    const int i0 = (int)(begin / 4), i1 = (int)(end / 4);
    int blocks = (i1 - i0 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;      // (block 0 carries the step's scalars even when the range is empty)
    Hyper nh;
    std::memset(&nh, 0, sizeof(nh));
    if (next) nh = *next;
    FeedPair fp;
    std::memset(&fp, 0, sizeof(fp));
    fp.T = g.T; fp.D = g.D; fp.pair_delta = g.pair_delta;
    for (int h = 0; h < 2; ++h) {
        const FeedJob &j = feeds[h];
        FeedHalf &f = fp.half[h];
        f.store = j.store; f.first_row = j.first_row; f.num_frames = j.num_frames; f.labels = j.labels; f.video_ids = j.video_ids;
        f.out = j.out; f.labels_out = j.labels_out; f.twin = j.twin; f.rows = j.n_videos > 0 ? j.n_videos * g.T : 0; f.bf16 = j.bf16;
    }
    const int rows = fp.half[0].rows + fp.half[1].rows;
    hipLaunchKernelGGL(sgd_open_feed_kernel, dim3(blocks + rows), dim3(256), 0, stream, g, params, grads, momentum, ws, i0, i1,
                       fused_norm ? g.o_sumsq : g.o_norm_part, fused_norm ? g.n_sumsq : g.n_norm_blocks, lr, mu, wd, clip, nh, next ? 1 : 0,
                       blocks, fp);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_sgd_fixup(const Geom &g, float *params, const float *grads, float *momentum, float *ws, float *p16, float lr, float mu,
                     float clip, const Hyper *next, hipStream_t stream) {
    Hyper nx;
    std::memset(&nx, 0, sizeof(nx));
    if (next) nx = *next;
    hipLaunchKernelGGL(sgd_fixup_kernel, dim3(128), dim3(256), 0, stream, g, params, grads, momentum, ws, p16, g.live_floats / 4, lr, mu, clip,
                       nx, next ? 1 : 0);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_to_bf16(const float *src, float *dst_twin, int64_t n, hipStream_t stream) {
    const int64_t n4 = n / 4;
    if (n4 <= 0) return 0;
    int blocks = (int)std::min<int64_t>((n4 + 255) / 256, 2048);
    hipLaunchKernelGGL(to_bf16_kernel, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const float4 *>(src),
                       reinterpret_cast<uint2 *>(dst_twin), n4);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_to_bf16_pair(const float *src, float *dst_hi, float *dst_lo, int64_t n, hipStream_t stream) {
    const int64_t n4 = n / 4;
    if (n4 <= 0) return 0;
    int blocks = (int)std::min<int64_t>((n4 + 255) / 256, 2048);
    hipLaunchKernelGGL(to_bf16_pair_kernel, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const float4 *>(src),
                       reinterpret_cast<uint2 *>(dst_hi), reinterpret_cast<uint2 *>(dst_lo), n4);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_set_hyper(float *ws_hyper, const Hyper &h, hipStream_t stream) {
    hipLaunchKernelGGL(set_hyper_kernel, dim3(1), dim3(64), 0, stream, reinterpret_cast<Hyper *>(ws_hyper), h);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_fill(float *dst, float value, int64_t n, hipStream_t stream) {
    int blocks = (int)((n + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(256), 0, stream, dst, value, n);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace ta3n
