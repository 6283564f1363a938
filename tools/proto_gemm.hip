// Standalone prototype / microbenchmark of the fp32-MFMA tile kernel structure
// (not part of the library): measures, on the GPU box,
//   (1) the v_mfma_f32_32x32x2_f32 issue-rate ceiling with every SIMD busy,
//   (2) LDS-DMA (global_load_lds_dwordx4) staged GEMM tiles for the three operand
//       combinations of the TA3N step at its real shapes:
//         NT  (A K-contiguous, B K-contiguous)  forward      X W^T
//         NN  (A K-contiguous, B k-major)        input grad   G W
//         TN  (A k-major,      B k-major)        weight grad  G^T X
// build: hipcc -O3 --offload-arch=gfx950 tools/proto_gemm.hip -o gpurun_out/proto_gemm
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

// ---------------------------------------------------------------- (1) MFMA ceiling
__global__ __launch_bounds__(256) void mfma_peak(float *out, int iters) {
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int i = 0; i < iters; ++i) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// ---------------------------------------------------------------- (2) tile GEMM

struct Gemm {
    const float *A, *B;
    float *C;
    int M, N, K, lda, ldb, ldc;
    const float *zeros;
};

// LDS-DMA issued from inline asm so that hipcc does not track it: a compiler-visible
// __builtin_amdgcn_global_load_lds makes hipcc put s_waitcnt vmcnt(0) in front of every
// later ds_read, which drains the prefetched stages.  Completion is counted by hand
// (s_waitcnt vmcnt(N) + s_barrier) in the main loop.  lds_byte_addr must be wave-uniform.
__device__ __forceinline__ void glds16(const float *gsrc, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}

// Stage image of a K-contiguous operand: [row][16 slots], slot' = slot ^ (row & 15).
// Stage image of a k-major operand:      [k][R] linear.
// One wave instruction moves 64 lanes x 16 B = 1 KiB to consecutive LDS bytes.
template <int R, bool KMAJOR, int BK, int NW>
__device__ __forceinline__ void issue_operand(const float *__restrict__ base, int ld, int r0, int rvalid, int k0, int klen,
                                              unsigned lds_op_addr, int wave, int lane, const float *__restrict__ zeros) {
    constexpr int NINSTR = R * BK * 4 / 1024;      // wave instructions per stage for this operand
    constexpr int PER_WAVE = NINSTR / NW;
    static_assert(NINSTR % NW == 0, "pieces must divide over the waves");
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
        const int q = wave + NW * i;               // instruction id -> 1 KiB piece q
        const float *src;
        if (!KMAJOR) {
            constexpr int SPR = BK / 4;            // 16-byte slots per row
            const int row = q * (64 / SPR) + lane / SPR;
            const int g = (lane % SPR) ^ (row & 15);
            const int k = k0 + 4 * g;
            src = (r0 + row < rvalid && k < klen) ? base + (size_t)(r0 + row) * ld + k : zeros;
        } else {
            constexpr int LPR = R / 4;             // lanes per k row
            constexpr int KPI = 64 / LPR;          // k rows per piece
            const int k = k0 + q * KPI + lane / LPR;
            const int r = r0 + (lane % LPR) * 4;
            src = (r < rvalid && k < klen) ? base + (size_t)k * ld + r : zeros;
        }
        glds16(src, lds_op_addr + q * 1024);
    }
}

template <int WM, int WN, int WK, int NBUF, bool AKM, bool BKM, int BK, int MODE = 0>
__global__ __launch_bounds__(64 * WM * WN * WK) void gemm_proto(Gemm g) {
    constexpr int NW = WM * WN * WK;
    constexpr int BM = 32 * WM, BN = 32 * WN;
    constexpr int STAGE = (BM + BN) * BK;          // floats
    constexpr int LPW = (BM + BN) * BK * 4 / 1024 / NW;   // glds instructions per wave per stage
    constexpr int GPW = BK / 4 / WK;               // 4-wide k groups per wave per stage
    constexpr int LDS_FLOATS = NBUF * STAGE > NW * 32 * 36 ? NBUF * STAGE : NW * 32 * 36;
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lptr_t *)lds);
    const int li = lane & 31, lh = lane >> 5;
    const int wk = wave % WK, wn = (wave / WK) % WN, wm = wave / (WK * WN);
    const int tiles_n = (g.N + BN - 1) / BN;
    const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
    const int nchunks = (g.K + BK - 1) / BK;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    auto issue = [&](int c) {
        const unsigned st = lds_base + (unsigned)((c % NBUF) * STAGE * 4);
        issue_operand<BM, AKM, BK, NW>(g.A, g.lda, m0, g.M, c * BK, g.K, st, wave, lane, g.zeros);
        issue_operand<BN, BKM, BK, NW>(g.B, g.ldb, n0, g.N, c * BK, g.K, st + BM * BK * 4, wave, lane, g.zeros);
    };
    // prologue: NBUF-1 stages in flight (stages past the end read zeros: keeps the vmcnt arithmetic uniform)
#pragma unroll
    for (int c = 0; c < NBUF - 1; ++c) issue(c);

    for (int c = 0; c < nchunks; ++c) {
        // this wave's pieces of stage c have landed once at most (NBUF-2) newer stages are outstanding
        if constexpr (NBUF == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if constexpr (NBUF == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPW) : "memory");
        __builtin_amdgcn_s_barrier();              // everyone's pieces landed; everyone finished reading stage c-1
        asm volatile("" ::: "memory");
        if (MODE != 1) issue(c + NBUF - 1);        // refill the buffer stage c-1 used (MODE 1: compute-only ceiling)
        const float *sa = lds + (c % NBUF) * STAGE;
        const float *sb = sa + BM * BK;
        if constexpr (MODE == 3 || MODE == 4) {
            // MODE 3: fp32 stages, operands rounded to bf16 in registers, bf16 MFMA.  MODE 4: the stage bytes ARE bf16
            // (timing probe of bf16 storage: K floats = 2K bf16, values meaningless).
            typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
            typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            static_assert(!AKM && !BKM, "bf16 probe: K-contiguous operands only");
            float4 ta[GPW / 2], tb[GPW / 2];
#pragma unroll
            for (int q = 0; q < GPW / 2; ++q) {
                const int G = wk * GPW + 2 * q + lh;
                const int ra = wm * 32 + li, rb = wn * 32 + li;
                ta[q] = *reinterpret_cast<const float4 *>(sa + ra * BK + ((G ^ (ra & 15)) << 2));
                tb[q] = *reinterpret_cast<const float4 *>(sb + rb * BK + ((G ^ (rb & 15)) << 2));
            }
            __builtin_amdgcn_sched_barrier(0);
            auto pk = [](float lo, float hi) { f32x2 v = {lo, hi}; return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)); };
            typedef short s16x4 __attribute__((ext_vector_type(4)));
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            if constexpr (MODE == 4) {
#pragma unroll
                for (int q = 0; q < GPW / 2; ++q)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ta[q]), __builtin_bit_cast(bf16x8, tb[q]), acc, 0, 0, 0);
            } else if constexpr ((GPW / 2) % 2 == 0) {
#pragma unroll
                for (int q = 0; q < GPW / 2; q += 2) {
                    const u32x4 a = {pk(ta[q].x, ta[q].y), pk(ta[q].z, ta[q].w), pk(ta[q + 1].x, ta[q + 1].y), pk(ta[q + 1].z, ta[q + 1].w)};
                    const u32x4 b = {pk(tb[q].x, tb[q].y), pk(tb[q].z, tb[q].w), pk(tb[q + 1].x, tb[q + 1].y), pk(tb[q + 1].z, tb[q + 1].w)};
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int q = 0; q < GPW / 2; ++q) {
                    const u32x2 a = {pk(ta[q].x, ta[q].y), pk(ta[q].z, ta[q].w)};
                    const u32x2 b = {pk(tb[q].x, tb[q].y), pk(tb[q].z, tb[q].w)};
                    acc = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), acc, 0, 0, 0);
                }
            }
            continue;
        }
#pragma unroll
        for (int q = 0; q < GPW / 2; ++q) {
            const int G = wk * GPW + 2 * q + lh;   // this half-wave's k group: k = 4G .. 4G+3
            float av[4], bv[4];
            if (!AKM) {
                const int r = wm * 32 + li;
                const float4 t = *reinterpret_cast<const float4 *>(sa + r * BK + ((G ^ (r & 15)) << 2));
                av[0] = t.x; av[1] = t.y; av[2] = t.z; av[3] = t.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) av[j] = sa[(4 * G + j) * BM + wm * 32 + li];
            }
            if (!BKM) {
                const int r = wn * 32 + li;
                const float4 t = *reinterpret_cast<const float4 *>(sb + r * BK + ((G ^ (r & 15)) << 2));
                bv[0] = t.x; bv[1] = t.y; bv[2] = t.z; bv[3] = t.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) bv[j] = sb[(4 * G + j) * BN + wn * 32 + li];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (MODE != 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc, 0, 0, 0);
                else acc[j] += av[j] * bv[j];      // MODE 2: DMA + LDS reads without the matrix pipe
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // epilogue: reduce the K split through LDS, store row-contiguous
    float *cs = lds + wave * (32 * 36);
#pragma unroll
    for (int r = 0; r < 16; ++r) cs[((r & 3) + 8 * (r >> 2) + 4 * lh) * 36 + li] = acc[r];
    __syncthreads();
    for (int idx = tid; idx < BM * BN / 4; idx += 64 * NW) {
        const int r = idx / (BN / 4), c4 = (idx % (BN / 4)) * 4;
        const int tile = (r >> 5) * WN + (c4 >> 5);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < WK; ++q) {
            const float4 p = *reinterpret_cast<const float4 *>(&lds[(tile * WK + q) * (32 * 36) + (r & 31) * 36 + (c4 & 31)]);
            v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        }
        const int m = m0 + r, n = n0 + c4;
        if (m < g.M && n < g.N) *reinterpret_cast<float4 *>(g.C + (size_t)m * g.ldc + n) = v;
    }
}

// ---------------------------------------------------------------- (3) private-pipeline variant
// Each wave of the workgroup streams its OWN K chunks (chunk c goes to wave c % NW) of the shared 32x32 tile
// through its own LDS stages: no s_barrier in the K loop, only the wave's own counted vmcnt; the NW partial
// accumulators are added through LDS once at the end.
template <int NW, int BKP, int NBUF, bool AKM, bool BKM>
__global__ __launch_bounds__(64 * NW) void gemm_priv(Gemm g) {
    constexpr int STAGE = 64 * BKP;                // floats per stage per wave: 32 A rows + 32 B rows
    constexpr int SPR = BKP / 4;                   // 16-byte slots per K-contiguous row
    constexpr int RPP = 64 / SPR;                  // rows per 1 KiB piece
    constexpr int PIECES = 32 / RPP;               // pieces per operand per stage (K-contiguous); k-major: 32*BKP*4/1024
    constexpr int PIECES_KM = 32 * BKP * 4 / 1024;
    constexpr int LPW = (AKM ? PIECES_KM : PIECES) + (BKM ? PIECES_KM : PIECES);
    constexpr int LDSF = NW * NBUF * STAGE > NW * 32 * 36 ? NW * NBUF * STAGE : NW * 32 * 36;
    __shared__ __attribute__((aligned(16))) float lds[LDSF];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lptr_t *)lds) + wave * NBUF * STAGE * 4;
    const float *my = lds + wave * NBUF * STAGE;
    const int li = lane & 31, lh = lane >> 5;
    const int tiles_n = (g.N + 31) / 32;
    const int m0 = (blockIdx.x / tiles_n) * 32, n0 = (blockIdx.x % tiles_n) * 32;
    const int nchunks = (g.K + BKP - 1) / BKP;
    const int mine = (nchunks - wave + NW - 1) / NW;            // chunks wave, wave + NW, ...
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    auto issue_op = [&](const float *base, int ld, bool kmajor, int r0, int rvalid, int k0, unsigned dst) {
        if (!kmajor) {
#pragma unroll
            for (int q = 0; q < PIECES; ++q) {
                const int row = q * RPP + lane / SPR;
                const int gq = (lane % SPR) ^ ((row >> (SPR == 8 ? 1 : 2)) & (SPR - 1));
                const int k = k0 + 4 * gq;
                const float *src = (r0 + row < rvalid && k < g.K) ? base + (size_t)(r0 + row) * ld + k : g.zeros;
                glds16(src, dst + q * 1024);
            }
        } else {
#pragma unroll
            for (int q = 0; q < PIECES_KM; ++q) {
                const int k = k0 + q * 8 + lane / 8;
                const int r = r0 + (lane % 8) * 4;
                const float *src = (r < rvalid && k < g.K) ? base + (size_t)k * ld + r : g.zeros;
                glds16(src, dst + q * 1024);
            }
        }
    };
    auto issue = [&](int i) {                       // i-th chunk of this wave
        const int k0 = (wave + i * NW) * BKP;
        const unsigned st = lds_base + (unsigned)((i % NBUF) * STAGE * 4);
        issue_op(g.A, g.lda, AKM, m0, g.M, k0, st);
        issue_op(g.B, g.ldb, BKM, n0, g.N, k0, st + 32 * BKP * 4);
    };
#pragma unroll
    for (int i = 0; i < NBUF - 1; ++i) issue(i);   // past-the-end chunks read zeros: keeps the vmcnt arithmetic uniform
    for (int i = 0; i < mine; ++i) {
        if constexpr (NBUF == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * LPW) : "memory");
        issue(i + NBUF - 1);
        const float *sa = my + (i % NBUF) * STAGE;
        const float *sb = sa + 32 * BKP;
        float av[BKP / 8][4], bv[BKP / 8][4];
#pragma unroll
        for (int q = 0; q < BKP / 8; ++q) {
            const int G = 2 * q + lh;
            if (!AKM) {
                const float4 t = *reinterpret_cast<const float4 *>(sa + li * BKP + ((G ^ ((li >> (SPR == 8 ? 1 : 2)) & (SPR - 1))) << 2));
                av[q][0] = t.x; av[q][1] = t.y; av[q][2] = t.z; av[q][3] = t.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) av[q][j] = sa[(4 * G + j) * 32 + li];
            }
            if (!BKM) {
                const float4 t = *reinterpret_cast<const float4 *>(sb + li * BKP + ((G ^ ((li >> (SPR == 8 ? 1 : 2)) & (SPR - 1))) << 2));
                bv[q][0] = t.x; bv[q][1] = t.y; bv[q][2] = t.z; bv[q][3] = t.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) bv[q][j] = sb[(4 * G + j) * 32 + li];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < BKP / 8; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][j], bv[q][j], acc, 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float *cs = lds + wave * (32 * 36);
#pragma unroll
    for (int r = 0; r < 16; ++r) cs[((r & 3) + 8 * (r >> 2) + 4 * lh) * 36 + li] = acc[r];
    __syncthreads();
    for (int idx = tid; idx < 32 * 32 / 4; idx += 64 * NW) {
        const int r = idx / 8, c4 = (idx % 8) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            const float4 p = *reinterpret_cast<const float4 *>(&lds[q * (32 * 36) + r * 36 + c4]);
            v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        }
        const int m = m0 + r, n = n0 + c4;
        if (m < g.M && n < g.N) *reinterpret_cast<float4 *>(g.C + (size_t)m * g.ldc + n) = v;
    }
}

// ---------------------------------------------------------------- (4) direct-to-register variant
// With the K split across the waves of a workgroup no two waves share operand data, so LDS staging buys
// nothing: each wave loads its own MFMA fragments straight from global memory (whole 128-byte lines per
// row and stage), double-buffered in registers; no LDS, no barrier and no DMA in the K loop.
// Stage = 32 k: lanes 0-31 hold k0..k0+15 of their row, lanes 32-63 hold k0+16..k0+31; MFMA j pairs k0+j with k0+16+j.
template <int NW, bool AKM, bool BKM>
__global__ __launch_bounds__(64 * NW) void gemm_direct(Gemm g) {
    __shared__ __attribute__((aligned(16))) float lds[NW * 32 * 36];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int tiles_n = (g.N + 31) / 32;
    const int m0 = (blockIdx.x / tiles_n) * 32, n0 = (blockIdx.x % tiles_n) * 32;
    const int nstages = (g.K + 31) / 32;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const bool a_ok = m0 + li < g.M, b_ok = n0 + li < g.N;
    auto load = [&](float (&a)[16], float (&b)[16], int s) {
        const int k0 = s * 32 + 16 * lh;
        if (!AKM) {
            const float *p = g.A + (size_t)(m0 + li) * g.lda + k0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 t = *reinterpret_cast<const float4 *>((a_ok && k0 + 4 * q < g.K) ? p + 4 * q : g.zeros);
                a[4 * q] = t.x; a[4 * q + 1] = t.y; a[4 * q + 2] = t.z; a[4 * q + 3] = t.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) a[j] = *((a_ok && k0 + j < g.K) ? g.A + (size_t)(k0 + j) * g.lda + m0 + li : g.zeros);
        }
        if (!BKM) {
            const float *p = g.B + (size_t)(n0 + li) * g.ldb + k0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 t = *reinterpret_cast<const float4 *>((b_ok && k0 + 4 * q < g.K) ? p + 4 * q : g.zeros);
                b[4 * q] = t.x; b[4 * q + 1] = t.y; b[4 * q + 2] = t.z; b[4 * q + 3] = t.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) b[j] = *((b_ok && k0 + j < g.K) ? g.B + (size_t)(k0 + j) * g.ldb + n0 + li : g.zeros);
        }
    };
    float a0[16], b0[16], a1[16], b1[16];
    int s = wave;
    load(a0, b0, s);
    for (; s < nstages; s += 2 * NW) {
        load(a1, b1, s + NW);                       // past-the-end stages read zeros
#pragma unroll
        for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc, 0, 0, 0);
        load(a0, b0, s + 2 * NW);
#pragma unroll
        for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc, 0, 0, 0);
    }
    float *cs = lds + wave * (32 * 36);
#pragma unroll
    for (int r = 0; r < 16; ++r) cs[((r & 3) + 8 * (r >> 2) + 4 * lh) * 36 + li] = acc[r];
    __syncthreads();
    for (int idx = tid; idx < 32 * 32 / 4; idx += 64 * NW) {
        const int r = idx / 8, c4 = (idx % 8) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            const float4 p = *reinterpret_cast<const float4 *>(&lds[q * (32 * 36) + r * 36 + c4]);
            v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        }
        const int m = m0 + r, n = n0 + c4;
        if (m < g.M && n < g.N) *reinterpret_cast<float4 *>(g.C + (size_t)m * g.ldc + n) = v;
    }
}

// ---------------------------------------------------------------- (5) loader-wave variant
// Wave specialisation: WK consumer waves (K split of one 32x32 tile) only read LDS and issue MFMAs; one extra wave
// issues every LDS-DMA of the workgroup, NBUF stages deep.  One s_barrier per chunk joins them.
template <int WK, int NBUF, bool AKM, bool BKM>
__global__ __launch_bounds__(64 * (WK + 1)) void gemm_loader(Gemm g) {
    constexpr int BK = 64;
    constexpr int STAGE = 64 * BK;                 // floats: 32 A rows + 32 B rows
    constexpr int LDSF = NBUF * STAGE > WK * 32 * 36 ? NBUF * STAGE : WK * 32 * 36;
    __shared__ __attribute__((aligned(16))) float lds[LDSF];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lptr_t *)lds);
    const int li = lane & 31, lh = lane >> 5;
    const int tiles_n = (g.N + 31) / 32;
    const int m0 = (blockIdx.x / tiles_n) * 32, n0 = (blockIdx.x % tiles_n) * 32;
    const int nchunks = (g.K + BK - 1) / BK;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (wave == WK) {
        // ---- loader: 8 pieces per operand per stage ----
        auto issue = [&](int c) {
            const unsigned st = lds_base + (unsigned)((c % NBUF) * STAGE * 4);
            const int k0 = c * BK;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float *src;
                if (!AKM) {
                    const int row = q * 4 + (lane >> 4);
                    const int k = k0 + 4 * ((lane & 15) ^ (row & 15));
                    src = (m0 + row < g.M && k < g.K) ? g.A + (size_t)(m0 + row) * g.lda + k : g.zeros;
                } else {
                    const int k = k0 + q * 8 + lane / 8;
                    const int r = m0 + (lane % 8) * 4;
                    src = (r < g.M && k < g.K) ? g.A + (size_t)k * g.lda + r : g.zeros;
                }
                glds16(src, st + q * 1024);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float *src;
                if (!BKM) {
                    const int row = q * 4 + (lane >> 4);
                    const int k = k0 + 4 * ((lane & 15) ^ (row & 15));
                    src = (n0 + row < g.N && k < g.K) ? g.B + (size_t)(n0 + row) * g.ldb + k : g.zeros;
                } else {
                    const int k = k0 + q * 8 + lane / 8;
                    const int r = n0 + (lane % 8) * 4;
                    src = (r < g.N && k < g.K) ? g.B + (size_t)k * g.ldb + r : g.zeros;
                }
                glds16(src, st + 32 * BK * 4 + q * 1024);
            }
        };
#pragma unroll
        for (int c = 0; c < NBUF - 1; ++c) issue(c);
        for (int c = 0; c < nchunks; ++c) {
            if constexpr (NBUF == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if constexpr (NBUF == 3) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            issue(c + NBUF - 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        const int wk = wave;
        constexpr int GPW = 16 / WK;
        for (int c = 0; c < nchunks; ++c) {
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const float *sa = lds + (c % NBUF) * STAGE;
            const float *sb = sa + 32 * BK;
            float av[GPW / 2][4], bv[GPW / 2][4];
#pragma unroll
            for (int q = 0; q < GPW / 2; ++q) {
                const int G = wk * GPW + 2 * q + lh;
                if (!AKM) {
                    const float4 t = *reinterpret_cast<const float4 *>(sa + li * BK + ((G ^ (li & 15)) << 2));
                    av[q][0] = t.x; av[q][1] = t.y; av[q][2] = t.z; av[q][3] = t.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) av[q][j] = sa[(4 * G + j) * 32 + li];
                }
                if (!BKM) {
                    const float4 t = *reinterpret_cast<const float4 *>(sb + li * BK + ((G ^ (li & 15)) << 2));
                    bv[q][0] = t.x; bv[q][1] = t.y; bv[q][2] = t.z; bv[q][3] = t.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) bv[q][j] = sb[(4 * G + j) * 32 + li];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < GPW / 2; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][j], bv[q][j], acc, 0, 0, 0);
        }
    }
    __syncthreads();
    if (wave < WK) {
        float *cs = lds + wave * (32 * 36);
#pragma unroll
        for (int r = 0; r < 16; ++r) cs[((r & 3) + 8 * (r >> 2) + 4 * lh) * 36 + li] = acc[r];
    }
    __syncthreads();
    for (int idx = tid; idx < 32 * 32 / 4; idx += 64 * (WK + 1)) {
        const int r = idx / 8, c4 = (idx % 8) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < WK; ++q) {
            const float4 p = *reinterpret_cast<const float4 *>(&lds[q * (32 * 36) + r * 36 + c4]);
            v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        }
        const int m = m0 + r, n = n0 + c4;
        if (m < g.M && n < g.N) *reinterpret_cast<float4 *>(g.C + (size_t)m * g.ldc + n) = v;
    }
}

static double cpu_ref(const std::vector<float> &A, const std::vector<float> &B, int lda, int ldb, bool akm, bool bkm, int K, int m, int n) {
    double s = 0;
    for (int k = 0; k < K; ++k) {
        const double a = akm ? A[(size_t)k * lda + m] : A[(size_t)m * lda + k];
        const double b = bkm ? B[(size_t)k * ldb + n] : B[(size_t)n * ldb + k];
        s += a * b;
    }
    return s;
}

template <int WM, int WN, int WK, int NBUF, bool AKM, bool BKM, int BK = 64, int MODE = 0>
void run(const char *name, int M, int N, int K) {
    const int lda = AKM ? M : K, ldb = BKM ? N : K;
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
    for (auto &v : hA) v = (float)rand() / (float)RAND_MAX - 0.5f;
    for (auto &v : hB) v = (float)rand() / (float)RAND_MAX - 0.5f;
    float *dA, *dB, *dC, *dz;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4)); CK(hipMalloc(&dz, 256));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dz, 0, 256));
    CK(hipMemset(dC, 0, (size_t)M * N * 4));
    Gemm g{dA, dB, dC, M, N, K, lda, ldb, N, dz};
    constexpr int BM = 32 * WM, BN = 32 * WN;
    const int grid = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((gemm_proto<WM, WN, WK, NBUF, AKM, BKM, BK, MODE>), dim3(grid), dim3(64 * WM * WN * WK), 0, 0, g);
    CK(hipDeviceSynchronize());
    const int reps = 50;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((gemm_proto<WM, WN, WK, NBUF, AKM, BKM, BK, MODE>), dim3(grid), dim3(64 * WM * WN * WK), 0, 0, g);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<float> hC((size_t)M * N);
    CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0;
    for (int t = 0; t < 400; ++t) {
        const int m = (t * 7919) % M, n = (t * 104729) % N;
        maxerr = fmax(maxerr, fabs(cpu_ref(hA, hB, lda, ldb, AKM, BKM, K, m, n) - hC[(size_t)m * N + n]));
    }
    // last row / col corners
    maxerr = fmax(maxerr, fabs(cpu_ref(hA, hB, lda, ldb, AKM, BKM, K, M - 1, N - 1) - hC[(size_t)(M - 1) * N + N - 1]));
    const double us = 1e3 * ms / reps;
    printf("%-12s %dx%dx%d mode%d tile %dx%d wk%d nbuf%d bk%d grid %5d : %8.2f us  %6.1f TF  maxerr %.2e\n", name, M, N, K, MODE, BM, BN, WK, NBUF, BK, grid, us,
           2.0 * M * N * K / us * 1e-6, maxerr);
    fflush(stdout);
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC)); CK(hipFree(dz));
}

template <int NW, int BKP, int NBUF, bool AKM, bool BKM>
void runp(const char *name, int M, int N, int K) {
    const int lda = AKM ? M : K, ldb = BKM ? N : K;
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
    for (auto &v : hA) v = (float)rand() / (float)RAND_MAX - 0.5f;
    for (auto &v : hB) v = (float)rand() / (float)RAND_MAX - 0.5f;
    float *dA, *dB, *dC, *dz;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4)); CK(hipMalloc(&dz, 256));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dz, 0, 256));
    CK(hipMemset(dC, 0, (size_t)M * N * 4));
    Gemm g{dA, dB, dC, M, N, K, lda, ldb, N, dz};
    const int grid = ((M + 31) / 32) * ((N + 31) / 32);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((gemm_priv<NW, BKP, NBUF, AKM, BKM>), dim3(grid), dim3(64 * NW), 0, 0, g);
    CK(hipDeviceSynchronize());
    const int reps = 50;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((gemm_priv<NW, BKP, NBUF, AKM, BKM>), dim3(grid), dim3(64 * NW), 0, 0, g);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<float> hC((size_t)M * N);
    CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0;
    for (int t = 0; t < 400; ++t) {
        const int m = (t * 7919) % M, n = (t * 104729) % N;
        maxerr = fmax(maxerr, fabs(cpu_ref(hA, hB, lda, ldb, AKM, BKM, K, m, n) - hC[(size_t)m * N + n]));
    }
    maxerr = fmax(maxerr, fabs(cpu_ref(hA, hB, lda, ldb, AKM, BKM, K, M - 1, N - 1) - hC[(size_t)(M - 1) * N + N - 1]));
    const double us = 1e3 * ms / reps;
    printf("%-12s %dx%dx%d PRIVATE nw%d bk%d nbuf%d grid %5d : %8.2f us  %6.1f TF  maxerr %.2e\n", name, M, N, K, NW, BKP, NBUF, grid, us,
           2.0 * M * N * K / us * 1e-6, maxerr);
    fflush(stdout);
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC)); CK(hipFree(dz));
}

template <int NW, bool AKM, bool BKM>
void rund(const char *name, int M, int N, int K) {
    const int lda = AKM ? M : K, ldb = BKM ? N : K;
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
    for (auto &v : hA) v = (float)rand() / (float)RAND_MAX - 0.5f;
    for (auto &v : hB) v = (float)rand() / (float)RAND_MAX - 0.5f;
    float *dA, *dB, *dC, *dz;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4)); CK(hipMalloc(&dz, 256));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dz, 0, 256));
    CK(hipMemset(dC, 0, (size_t)M * N * 4));
    Gemm g{dA, dB, dC, M, N, K, lda, ldb, N, dz};
    const int grid = ((M + 31) / 32) * ((N + 31) / 32);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((gemm_direct<NW, AKM, BKM>), dim3(grid), dim3(64 * NW), 0, 0, g);
    CK(hipDeviceSynchronize());
    const int reps = 50;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((gemm_direct<NW, AKM, BKM>), dim3(grid), dim3(64 * NW), 0, 0, g);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<float> hC((size_t)M * N);
    CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0;
    for (int t = 0; t < 400; ++t) {
        const int m = (t * 7919) % M, n = (t * 104729) % N;
        maxerr = fmax(maxerr, fabs(cpu_ref(hA, hB, lda, ldb, AKM, BKM, K, m, n) - hC[(size_t)m * N + n]));
    }
    maxerr = fmax(maxerr, fabs(cpu_ref(hA, hB, lda, ldb, AKM, BKM, K, M - 1, N - 1) - hC[(size_t)(M - 1) * N + N - 1]));
    const double us = 1e3 * ms / reps;
    printf("%-12s %dx%dx%d DIRECT nw%d grid %5d : %8.2f us  %6.1f TF  maxerr %.2e\n", name, M, N, K, NW, grid, us,
           2.0 * M * N * K / us * 1e-6, maxerr);
    fflush(stdout);
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC)); CK(hipFree(dz));
}

template <int WK, int NBUF, bool AKM, bool BKM>
void runl(const char *name, int M, int N, int K) {
    const int lda = AKM ? M : K, ldb = BKM ? N : K;
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
    for (auto &v : hA) v = (float)rand() / (float)RAND_MAX - 0.5f;
    for (auto &v : hB) v = (float)rand() / (float)RAND_MAX - 0.5f;
    float *dA, *dB, *dC, *dz;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4)); CK(hipMalloc(&dz, 256));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dz, 0, 256));
    CK(hipMemset(dC, 0, (size_t)M * N * 4));
    Gemm g{dA, dB, dC, M, N, K, lda, ldb, N, dz};
    const int grid = ((M + 31) / 32) * ((N + 31) / 32);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((gemm_loader<WK, NBUF, AKM, BKM>), dim3(grid), dim3(64 * (WK + 1)), 0, 0, g);
    CK(hipDeviceSynchronize());
    const int reps = 50;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((gemm_loader<WK, NBUF, AKM, BKM>), dim3(grid), dim3(64 * (WK + 1)), 0, 0, g);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<float> hC((size_t)M * N);
    CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0;
    for (int t = 0; t < 400; ++t) {
        const int m = (t * 7919) % M, n = (t * 104729) % N;
        maxerr = fmax(maxerr, fabs(cpu_ref(hA, hB, lda, ldb, AKM, BKM, K, m, n) - hC[(size_t)m * N + n]));
    }
    maxerr = fmax(maxerr, fabs(cpu_ref(hA, hB, lda, ldb, AKM, BKM, K, M - 1, N - 1) - hC[(size_t)(M - 1) * N + N - 1]));
    const double us = 1e3 * ms / reps;
    printf("%-12s %dx%dx%d LOADER wk%d nbuf%d grid %5d : %8.2f us  %6.1f TF  maxerr %.2e\n", name, M, N, K, WK, NBUF, grid, us,
           2.0 * M * N * K / us * 1e-6, maxerr);
    fflush(stdout);
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC)); CK(hipFree(dz));
}

int main() {
    {   // (1) ceiling
        float *out;
        CK(hipMalloc(&out, 2048 * 256 * 4));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int blocks : {256, 512, 1024, 2048}) {
            const int iters = 4096;
            hipLaunchKernelGGL(mfma_peak, dim3(blocks), dim3(256), 0, 0, out, iters);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(mfma_peak, dim3(blocks), dim3(256), 0, 0, out, iters);
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double flops = (double)blocks * 4 * iters * 2 * 4096.0;
            printf("mfma_peak blocks %4d: %.1f us  %.1f TF\n", blocks, ms * 1e3, flops / ms * 1e-9);
        }
        CK(hipFree(out));
    }
    // bf16-MFMA probes on the F1 shape: fp32 stages + in-register rounding (mode 3), bf16 stage bytes (mode 4, K halves)
#define SWEEP(WM, WN, WK)                                                       \
    run<WM, WN, WK, 2, false, false, 64, 3>("F1 cvt", 1010, 512, 2048);         \
    run<WM, WN, WK, 3, false, false, 64, 3>("F1 cvt", 1010, 512, 2048);         \
    run<WM, WN, WK, 4, false, false, 64, 3>("F1 cvt", 1010, 512, 2048);         \
    run<WM, WN, WK, 2, false, false, 128, 3>("F1 cvt", 1010, 512, 2048);        \
    run<WM, WN, WK, 2, false, false, 64, 4>("F1 bf16st", 1010, 512, 1024);      \
    run<WM, WN, WK, 3, false, false, 64, 4>("F1 bf16st", 1010, 512, 1024);      \
    run<WM, WN, WK, 4, false, false, 64, 4>("F1 bf16st", 1010, 512, 1024);      \
    run<WM, WN, WK, 2, false, false, 128, 4>("F1 bf16st", 1010, 512, 1024);
    SWEEP(1, 1, 4)
    SWEEP(1, 2, 4)
    SWEEP(2, 2, 2)
    run<1, 1, 8, 2, false, false, 64, 3>("F1 cvt", 1010, 512, 2048);
    run<1, 1, 8, 4, false, false, 64, 3>("F1 cvt", 1010, 512, 2048);
    run<1, 1, 8, 2, false, false, 64, 4>("F1 bf16st", 1010, 512, 1024);
    run<1, 1, 8, 4, false, false, 64, 4>("F1 bf16st", 1010, 512, 1024);
    return 0;
}
