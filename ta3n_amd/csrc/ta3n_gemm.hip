// Tile-list fp32 GEMM for gfx950 (CDNA4) on the exact-f32 matrix cores
// (v_mfma_f32_32x32x2_f32: 64 cycles/SIMD, bitwise an fmaf chain, 157 TF peak).
//
// One launch = one dependency level of the TA3N train step.  Each workgroup
// (256 threads = 4 wave64) takes one Task: a (32*WM x 32*WN) output tile whose
// K loop runs over a list of Segs.  A Seg is an affine view
//     A(r,k) = base_a[a_off + (kmajor ? k*a_ld + r : r*a_ld + k)]
// so the same kernel does  X W^T (forward), G W (input gradients), G^T X (weight
// gradients), the TRN frame-tuple gather+concat (one Seg per tuple position,
// reference TRNmodule.py:60-63/75-77 - never materialised), the scatter-free
// TRN input gradient (one Seg per (tuple,position) that contains the frame) and
// GradReverse (reference models.py:20-29) as a "scale the accumulator by -beta
// after this Seg" flag.  WK > 1 splits every 64-deep K chunk across the
// workgroup's waves so small outputs still occupy all 4 SIMDs of a CU.
//
// Data path per 64-deep chunk: global -> registers (float4, coalesced along the
// operand's contiguous axis) -> LDS in k-major form [k][row] (K-contiguous
// operands are transposed on the way in, row stride R+1 => <=2-way write
// conflicts; k-major operands are stored as-is with ds_write_b128) -> one
// ds_read_b32 per operand per MFMA (lanes 0-31 read 32 consecutive floats, the
// two half-waves hit different k rows: conflict free).  LDS is double buffered:
// one barrier per chunk, next chunk's global loads in flight during the MFMAs.
// The epilogue goes through LDS once more so the K-split partials are reduced
// and the stores / bias / mask operands are row-contiguous float4s.
#include <hip/hip_runtime.h>

#include "ta3n_kernels.h"

using namespace ta3n;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BKC = 64;        // K chunk staged per barrier
constexpr int NTHREADS = 256;

__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// rows x 64 chunk of one operand -> registers.  R = rows of the tile (32 or 64).
// Branch-free AND select-free per lane: an out-of-range element is read from a
// 16-byte block of zeros in the workspace (the validity test selects the ADDRESS,
// not the loaded value).  The loaded registers are therefore first touched by the
// LDS write of the NEXT iteration, so the loads stay in flight across the MFMAs of
// the current chunk.  (A per-element "if (valid) load" compiles to an exec-masked
// branch with its own s_waitcnt vmcnt(0); a select on the loaded value makes hipcc
// wait for each load right after issuing it.)  The only branch is wave-uniform:
// float4 path when the operand is 16-byte tileable.
template <int R>
__device__ __forceinline__ void g2r(float4 (&v)[R / 16], const float *__restrict__ base, int off, int ld, int kmajor,
                                    int r0, int rvalid, int k0, int klen, int tid, const float *__restrict__ zeros) {
    const float *__restrict__ origin = base + (size_t)off;
    if (!kmajor) {
        // element (row, k) at row*ld + k ; this lane: 4 consecutive k of R/16 rows
        const bool vec = ((off | ld | klen) & 3) == 0;
        const int k = k0 + (tid & 15) * 4;
        if (vec) {
            const bool kin = k < klen;              // klen % 4 == 0: the float4 is all-in or all-out
#pragma unroll
            for (int i = 0; i < R / 16; ++i) {
                const int row = r0 + (tid >> 4) + 16 * i;
                const float *p = (kin && row < rvalid) ? origin + (size_t)row * ld + k : zeros;
                v[i] = *reinterpret_cast<const float4 *>(p);
            }
        } else {
#pragma unroll
            for (int i = 0; i < R / 16; ++i) {
                const int row = r0 + (tid >> 4) + 16 * i;
                const float *pr = origin + (size_t)row * ld + k;
                const bool rok = row < rvalid;
                v[i].x = *((rok && k + 0 < klen) ? pr + 0 : zeros);
                v[i].y = *((rok && k + 1 < klen) ? pr + 1 : zeros);
                v[i].z = *((rok && k + 2 < klen) ? pr + 2 : zeros);
                v[i].w = *((rok && k + 3 < klen) ? pr + 3 : zeros);
            }
        }
    } else {
        // element (r, k) at k*ld + r ; this lane: 4 consecutive r of R/16 k-rows
        constexpr int TPR = R / 4;          // threads per k row
        constexpr int KPP = NTHREADS / TPR;  // k rows per pass
        const bool vec = ((off | ld | rvalid) & 3) == 0;
        const int col = r0 + (tid % TPR) * 4;
        if (vec) {
            const bool cin = col < rvalid;          // rvalid % 4 == 0
#pragma unroll
            for (int i = 0; i < R / 16; ++i) {
                const int k = k0 + tid / TPR + KPP * i;
                const float *p = (cin && k < klen) ? origin + (size_t)k * ld + col : zeros;
                v[i] = *reinterpret_cast<const float4 *>(p);
            }
        } else {
#pragma unroll
            for (int i = 0; i < R / 16; ++i) {
                const int k = k0 + tid / TPR + KPP * i;
                const float *pr = origin + (size_t)k * ld + col;
                const bool kok = k < klen;
                v[i].x = *((kok && col + 0 < rvalid) ? pr + 0 : zeros);
                v[i].y = *((kok && col + 1 < rvalid) ? pr + 1 : zeros);
                v[i].z = *((kok && col + 2 < rvalid) ? pr + 2 : zeros);
                v[i].w = *((kok && col + 3 < rvalid) ? pr + 3 : zeros);
            }
        }
    }
}

// registers -> LDS, k-major image [k][row] with row stride R+1 (transposing
// path) or R+4 (straight path).
template <int R>
__device__ __forceinline__ void r2s(const float4 (&v)[R / 16], float *__restrict__ s, int kmajor, int tid) {
    if (!kmajor) {
        constexpr int S = R + 1;
        const int kq = (tid & 15) * 4;
#pragma unroll
        for (int i = 0; i < R / 16; ++i) {
            const int row = (tid >> 4) + 16 * i;
            s[(kq + 0) * S + row] = v[i].x;
            s[(kq + 1) * S + row] = v[i].y;
            s[(kq + 2) * S + row] = v[i].z;
            s[(kq + 3) * S + row] = v[i].w;
        }
    } else {
        constexpr int S = R + 4;
        constexpr int TPR = R / 4;
        constexpr int KPP = NTHREADS / TPR;
        const int c = (tid % TPR) * 4;
#pragma unroll
        for (int i = 0; i < R / 16; ++i) {
            const int k = tid / TPR + KPP * i;
            *reinterpret_cast<float4 *>(&s[k * S + c]) = v[i];
        }
    }
}

__device__ __forceinline__ const float *base_ptr(const Ptrs &p, int base) {
    switch (base) {
        case BASE_X: return p.x;
        case BASE_P: return p.p;
        case BASE_G: return p.g;
        default: return p.ws;
    }
}

}  // namespace

namespace ta3n {

template <int WM, int WN, int WK>
__global__ __launch_bounds__(NTHREADS) void gemm_tiles(const Task *__restrict__ tasks, const Seg *__restrict__ segs,
                                                        Ptrs ptrs, int hyper_off, int zeros_off) {
    constexpr int BM = 32 * WM, BN = 32 * WN;
    constexpr int KW = BKC / WK;                 // k per wave per chunk
    constexpr int LA = BKC * (BM + 4), LB = BKC * (BN + 4);
    constexpr int BUF = LA + LB;
    static_assert(2 * BUF >= 4 * 32 * 36, "epilogue staging must fit");
    __shared__ __attribute__((aligned(16))) float lds[2 * BUF];

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int li = lane & 31, lh = lane >> 5;
    const int wk = wave % WK, wn = (wave / WK) % WN, wm = wave / (WK * WN);

    const Task &t = tasks[blockIdx.x];
    if (t.seg_count == 0) return;   // padding task of the XCD-aware ordering (uniform for the workgroup)
    const Hyper *__restrict__ hy = reinterpret_cast<const Hyper *>(ptrs.ws + hyper_off);
    const float *__restrict__ zeros = ptrs.ws + zeros_off;   // 64 floats that are never written

    float4 ra[BM / 16], rb[BN / 16];
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    int seg = 0, k0 = 0;
    Seg s = segs[t.seg_begin];
    g2r<BM>(ra, base_ptr(ptrs, s.a_base), s.a_off, s.a_ld, s.a_kmajor, t.m0, t.m_valid, k0, s.klen, tid, zeros);
    g2r<BN>(rb, base_ptr(ptrs, s.b_base), s.b_off, s.b_ld, s.b_kmajor, t.n0, t.n_valid, k0, s.klen, tid, zeros);
    int buf = 0;
    for (;;) {
        float *sa = lds + buf * BUF;
        float *sb = sa + LA;
        r2s<BM>(ra, sa, s.a_kmajor, tid);
        r2s<BN>(rb, sb, s.b_kmajor, tid);
        __syncthreads();
        // what the compute step of this chunk needs
        const int stride_a = s.a_kmajor ? BM + 4 : BM + 1;
        const int stride_b = s.b_kmajor ? BN + 4 : BN + 1;
        const int krem = s.klen - k0;                 // valid k in this chunk (may exceed 64)
        const bool seg_done = krem <= BKC;
        const int scale_kind = s.scale_kind;
        // advance and prefetch the next chunk into registers
        bool more = true;
        if (!seg_done) {
            k0 += BKC;
        } else {
            ++seg;
            k0 = 0;
            if (seg < t.seg_count) s = segs[t.seg_begin + seg];
            else more = false;
        }
        if (more) {
            g2r<BM>(ra, base_ptr(ptrs, s.a_base), s.a_off, s.a_ld, s.a_kmajor, t.m0, t.m_valid, k0, s.klen, tid, zeros);
            g2r<BN>(rb, base_ptr(ptrs, s.b_base), s.b_off, s.b_ld, s.b_kmajor, t.n0, t.n_valid, k0, s.klen, tid, zeros);
        }
        // MFMA over this wave's K slice of the chunk, 8 k (4 MFMAs) per group
        const float *pa = sa + (wk * KW + lh) * stride_a + wm * 32 + li;
        const float *pb = sb + (wk * KW + lh) * stride_b + wn * 32 + li;
#pragma unroll
        for (int g = 0; g < KW; g += 8) {
            if (wk * KW + g < krem) {
#pragma unroll
                for (int kk = 0; kk < 8; kk += 2) {
                    const float a = pa[(g + kk) * stride_a];
                    const float b = pb[(g + kk) * stride_b];
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
                }
            }
        }
        if (seg_done && scale_kind != SK_ONE) {
            const float sc = hyper_scale(hy, scale_kind);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] *= sc;
        }
        if (!more) break;
        buf ^= 1;
    }

    // ---- epilogue: accumulators -> LDS (reduces the K split, makes rows contiguous) ----
    __syncthreads();
    {
        float *cs = lds + wave * (32 * 36);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;   // 32x32 C/D fragment layout
            cs[row * 36 + li] = acc[r];
        }
    }
    __syncthreads();

    const uint32_t epi = t.epi;
    const float alpha = hyper_scale(hy, t.alpha_kind);
    const float gamma = hyper_scale(hy, t.gamma_kind);
    const bool drop_on = (epi & (EPI_DROP_I | EPI_DROP_V)) && hy->train != 0;
    const uint32_t dseed = (epi & EPI_DROP_I) ? hy->seed_i : hy->seed_v;
    const float dp = (epi & EPI_DROP_I) ? hy->p_drop_i : hy->p_drop_v;
    float *__restrict__ cbase = const_cast<float *>(base_ptr(ptrs, t.c_base)) + (size_t)t.c_off;
    // Operand reads are address-selected (absent / out-of-range -> a block of zeros, or of
    // ones for the mask) so that all of a thread's epilogue loads issue together.
    const float *__restrict__ ones = zeros + 64;   // region "ones" follows region "zeros" (plan builder)
    const float *__restrict__ bias = (epi & EPI_BIAS) ? base_ptr(ptrs, t.bias_base) + t.bias_off : nullptr;
    const float *__restrict__ aux = (epi & EPI_MASK) ? base_ptr(ptrs, t.aux_base) + t.aux_off : nullptr;
    const float *__restrict__ add = (epi & EPI_ADD) ? base_ptr(ptrs, t.add_base) + t.add_off : nullptr;
    const bool c_vec = ((t.c_off | t.c_ld) & 3) == 0;
    const int nfan = t.fan_count;

    for (int idx = tid; idx < BM * BN / 4; idx += NTHREADS) {
        const int r = idx / (BN / 4);
        const int c4 = (idx % (BN / 4)) * 4;
        const int tile = (r >> 5) * WN + (c4 >> 5);
        float4 v4 = zero4();
#pragma unroll
        for (int q = 0; q < WK; ++q) {
            const float4 part = *reinterpret_cast<const float4 *>(&lds[(tile * WK + q) * (32 * 36) + (r & 31) * 36 + (c4 & 31)]);
            v4.x += part.x; v4.y += part.y; v4.z += part.z; v4.w += part.w;
        }
        const int m = t.m0 + r, n = t.n0 + c4;
        const bool row_ok = m < t.m_valid;
        const int nrem = row_ok ? t.n_valid - n : 0;   // number of valid columns of this float4 (<= 0: none)
        float v[4] = {v4.x, v4.y, v4.z, v4.w};
        float bv[4], av[4], mv[4], fm[3][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool ok = e < nrem;
            bv[e] = *((ok && bias) ? bias + n + e : zeros);
            av[e] = *((ok && add) ? add + (size_t)m * t.add_ld + n + e : zeros);
            mv[e] = *((ok && aux) ? aux + (size_t)m * t.aux_ld + n + e : ones);
#pragma unroll
            for (int f = 0; f < 3; ++f)
                fm[f][e] = *((ok && f < nfan) ? ptrs.ws + (size_t)t.fan_mask_off[f] + (size_t)m * t.fan_ld + n + e : zeros);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x = (v[e] + bv[e]) * alpha + av[e];
            if (epi & EPI_RELU) x = fmaxf(x, 0.f);
            x = mv[e] > 0.f ? x : 0.f;
            if (drop_on) x *= keep_mask(dseed, (uint32_t)(m * t.drop_ld + n + e), dp);
            v[e] = x * gamma;
        }
        if (nrem <= 0) continue;
        float *cp = cbase + (size_t)m * t.c_ld + n;
        if (nrem >= 4 && c_vec) {
            *reinterpret_cast<float4 *>(cp) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (e < nrem) cp[e] = v[e];
        }
#pragma unroll
        for (int f = 0; f < 3; ++f) {   // same value through several ReLU masks (TRN tuples of one scale)
            if (f < nfan) {
                float *op = ptrs.ws + (size_t)t.fan_out_off[f] + (size_t)m * t.fan_ld + n;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (e < nrem) op[e] = fm[f][e] > 0.f ? v[e] : 0.f;
            }
        }
    }
}

template __global__ void gemm_tiles<1, 1, 4>(const Task *, const Seg *, Ptrs, int, int);
template __global__ void gemm_tiles<2, 1, 2>(const Task *, const Seg *, Ptrs, int, int);
template __global__ void gemm_tiles<1, 2, 2>(const Task *, const Seg *, Ptrs, int, int);
template __global__ void gemm_tiles<2, 2, 1>(const Task *, const Seg *, Ptrs, int, int);

int launch_gemm(const Phase &ph, const Task *d_tasks, const Seg *d_segs, const Ptrs &ptrs, int hyper_off,
                int zeros_off, hipStream_t stream) {
    if (ph.task_count == 0) return 0;
    const dim3 grid(ph.task_count), block(NTHREADS);
    const Task *tp = d_tasks + ph.task_begin;
    const int cfg = ph.wm * 100 + ph.wn * 10 + ph.wk;
    switch (cfg) {
        case 114: hipLaunchKernelGGL((gemm_tiles<1, 1, 4>), grid, block, 0, stream, tp, d_segs, ptrs, hyper_off, zeros_off); break;
        case 212: hipLaunchKernelGGL((gemm_tiles<2, 1, 2>), grid, block, 0, stream, tp, d_segs, ptrs, hyper_off, zeros_off); break;
        case 122: hipLaunchKernelGGL((gemm_tiles<1, 2, 2>), grid, block, 0, stream, tp, d_segs, ptrs, hyper_off, zeros_off); break;
        case 221: hipLaunchKernelGGL((gemm_tiles<2, 2, 1>), grid, block, 0, stream, tp, d_segs, ptrs, hyper_off, zeros_off); break;
        default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace ta3n
