// Standalone microbenchmark (not part of the library): where does the time of a small bf16 tile GEMM go on gfx950?
// Shape = the shared frame FC of the TA3N step, X[1010,2048] W[512,2048]^T, both operands bf16 and K-contiguous,
// LDS-DMA staged exactly like ta3n::gemm_tiles<.., MODE 2>.  Knobs (template parameters):
//   NBUF   LDS stages in flight
//   PF     0: none; 1: at kernel start every lane touches one 128-B line of this workgroup's operand panels per
//          instruction (global_load_dword, result discarded) - pulls the tile's first-touch traffic into this XCD's L2
//          at full memory-level parallelism instead of at the pace of the K loop
//   PASSES run the K loop twice: the second pass reads an L2-warm working set (ceiling of the loop structure)
// Per-workgroup s_memtime stamps are written out so the split prologue / loop / epilogue can be read.
// build: hipcc -O3 --offload-arch=gfx950 tools/proto_bf16.hip -o gpurun_out/proto_bf16
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lptr_t;

struct Gemm {
    const unsigned short *A, *B;   // bf16 [M][K], [N][K]
    float *C;
    int M, N, K, ld;               // ld: row stride of A and B in elements (>= K)
    unsigned long long *stamps;    // [grid][4]
    const unsigned short *zeros;
};

__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}

template <int NW>
__device__ __forceinline__ void wait_vm(int n);

template <int WM, int WN, int WK, int NBUF, int PF, int PASSES>
__global__ __launch_bounds__(64 * WM * WN * WK) void gemm_bf16(Gemm g) {
    constexpr int NW = WM * WN * WK;
    constexpr int BM = 32 * WM, BN = 32 * WN, BK = 128;        // 128 bf16 = 256 B per row and stage
    constexpr int STAGE_B = (BM + BN) * 256;                   // bytes
    constexpr int PA = BM / 4 / NW > 0 ? BM / 4 / NW : 1, PB = BN / 4 / NW > 0 ? BN / 4 / NW : 1;
    static_assert((BM / 4) % NW == 0 && (BN / 4) % NW == 0, "pieces must divide over the waves");
    constexpr int LPW = PA + PB;
    constexpr int EPI_B = NW * 32 * 36 * 4;
    constexpr int LDS_B = NBUF * STAGE_B > EPI_B ? NBUF * STAGE_B : EPI_B;
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_B + 256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lptr_t *)lds);
    const int li = lane & 31, lh = lane >> 5;
    const int wk = wave % WK, wn = (wave / WK) % WN, wm = wave / (WK * WN);
    // XCD-aware: workgroup b runs on XCD b % 8; give an XCD consecutive row panels
    const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
    const int nwg = tiles_m * tiles_n;
    const int xcd = blockIdx.x % 8, slot = blockIdx.x / 8;
    const int per = (nwg + 7) / 8;
    const int id = xcd * per + slot;
    if (id >= nwg) return;
    const int m0 = (id / tiles_n) * BM, n0 = (id % tiles_n) * BN;
    const int nchunks = g.K / BK;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();

    if (PF == 1) {   // touch every 128-B line of this tile's panels once: rows x (K*2/128) lines; 4-byte LDS-DMA into a dummy
                     // LDS word per lane (no VGPR destination to protect), 64 distinct lines per wave instruction
        const int lines_per_row = g.K * 2 / 128;
        const int total = (BM + BN) * lines_per_row;
        for (int i = tid; i < total; i += 64 * NW) {
            const int row = i / lines_per_row, ln = i % lines_per_row;
            const unsigned short *p = row < BM ? g.A + (size_t)min(m0 + row, g.M - 1) * g.ld + ln * 64
                                               : g.B + (size_t)min(n0 + row - BM, g.N - 1) * g.ld + ln * 64;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(p), "s"(lds_base + (unsigned)LDS_B) : "memory");
        }
    }

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    // per-lane DMA sources
    const unsigned char *pa[PA], *pb[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int q = wave + NW * i, row = q * 4 + (lane >> 4), s = (lane & 15) ^ (row & 15);
        pa[i] = (m0 + row < g.M) ? reinterpret_cast<const unsigned char *>(g.A + (size_t)(m0 + row) * g.ld) + 16 * s
                                 : reinterpret_cast<const unsigned char *>(g.zeros);
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int q = wave + NW * i, row = q * 4 + (lane >> 4), s = (lane & 15) ^ (row & 15);
        pb[i] = (n0 + row < g.N) ? reinterpret_cast<const unsigned char *>(g.B + (size_t)(n0 + row) * g.ld) + 16 * s
                                 : reinterpret_cast<const unsigned char *>(g.zeros);
    }
    const bool a_ok[1] = {true};
    (void)a_ok;
    auto issue = [&](int c, int buf) {
        const unsigned st = lds_base + (unsigned)(buf * STAGE_B);
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const bool ok = (m0 + (wave + NW * i) * 4 + (lane >> 4)) < g.M;
            glds16(ok ? pa[i] + (size_t)c * 256 : pa[i], st + (wave + NW * i) * 1024);
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const bool ok = (n0 + (wave + NW * i) * 4 + (lane >> 4)) < g.N;
            glds16(ok ? pb[i] + (size_t)c * 256 : pb[i], st + BM * 256 + (wave + NW * i) * 1024);
        }
    };
    unsigned long long t1 = 0, t2 = 0;
    const int ra = wm * 32 + li, rb = wn * 32 + li;
    for (int pass = 0; pass < PASSES; ++pass) {
        if (pass == 1) t2 = __builtin_amdgcn_s_memtime();
        int ibuf = 0;
#pragma unroll 1
        for (int c = 0; c < NBUF - 1; ++c) { issue(c < nchunks ? c : nchunks - 1, ibuf); ibuf = ibuf + 1 == NBUF ? 0 : ibuf + 1; }
        int cbuf = 0;
        for (int c = 0; c < nchunks; ++c) {
            if constexpr (NBUF == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * LPW) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (c == 0 && pass == 0) t1 = __builtin_amdgcn_s_memtime();
            { const int cn = c + NBUF - 1; issue(cn < nchunks ? cn : nchunks - 1, ibuf); ibuf = ibuf + 1 == NBUF ? 0 : ibuf + 1; }
            const unsigned char *sa = lds + cbuf * STAGE_B, *sb = sa + BM * 256;
            constexpr int SPW = 16 / WK / 2;   // MFMAs (16 k each) per wave per stage
            u32x4 fa[SPW], fb[SPW];
#pragma unroll
            for (int q = 0; q < SPW; ++q) {
                const int G = (wk * SPW + q) * 2 + lh;
                fa[q] = *reinterpret_cast<const u32x4 *>(sa + ra * 256 + ((G ^ (ra & 15)) << 4));
                fb[q] = *reinterpret_cast<const u32x4 *>(sb + rb * 256 + ((G ^ (rb & 15)) << 4));
            }
#pragma unroll
            for (int q = 0; q < SPW; ++q)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[q]), __builtin_bit_cast(bf16x8, fb[q]), acc, 0, 0, 0);
            cbuf = cbuf + 1 == NBUF ? 0 : cbuf + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    unsigned long long t3 = __builtin_amdgcn_s_memtime();
    // epilogue through LDS (K-split reduction), float4 stores
    float *cs = reinterpret_cast<float *>(lds) + wave * (32 * 36);
#pragma unroll
    for (int r = 0; r < 16; ++r) cs[((r & 3) + 8 * (r >> 2) + 4 * lh) * 36 + li] = acc[r];
    __syncthreads();
    const float inv = 1.f / PASSES;
    for (int idx = tid; idx < BM * BN / 4; idx += 64 * NW) {
        const int r = idx / (BN / 4), c4 = (idx % (BN / 4)) * 4;
        const int tile = (r >> 5) * WN + (c4 >> 5);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < WK; ++q) {
            const float4 p = *reinterpret_cast<const float4 *>(reinterpret_cast<float *>(lds) + (tile * WK + q) * (32 * 36) + (r & 31) * 36 + (c4 & 31));
            v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        }
        if (m0 + r < g.M && n0 + c4 < g.N)
            *reinterpret_cast<float4 *>(g.C + (size_t)(m0 + r) * g.N + n0 + c4) = make_float4(v.x * inv, v.y * inv, v.z * inv, v.w * inv);
    }
    if (tid == 0) {
        unsigned long long t4 = __builtin_amdgcn_s_memtime();
        g.stamps[blockIdx.x * 4 + 0] = t1 - t0;
        g.stamps[blockIdx.x * 4 + 1] = (PASSES > 1 ? t2 : t3) - t1;
        g.stamps[blockIdx.x * 4 + 2] = PASSES > 1 ? t3 - t2 : 0;
        g.stamps[blockIdx.x * 4 + 3] = t4 - t3;
    }
}

static unsigned short f2bf(float f) {
    unsigned u; memcpy(&u, &f, 4);
    u += 0x7FFF + ((u >> 16) & 1);
    return (unsigned short)(u >> 16);
}
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

template <int WM, int WN, int WK, int NBUF, int PF, int PASSES>
void run(const char *label, Gemm g, const std::vector<float> &ref, int reps) {
    constexpr int BM = 32 * WM, BN = 32 * WN;
    const int nwg = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    const int grid = (nwg + 7) / 8 * 8;
    CK(hipMemset(g.C, 0, (size_t)g.M * g.N * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gemm_bf16<WM, WN, WK, NBUF, PF, PASSES>), dim3(grid), dim3(64 * WM * WN * WK), 0, 0, g);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((gemm_bf16<WM, WN, WK, NBUF, PF, PASSES>), dim3(grid), dim3(64 * WM * WN * WK), 0, 0, g);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<float> c((size_t)g.M * g.N);
    CK(hipMemcpy(c.data(), g.C, c.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0;
    for (size_t i = 0; i < c.size(); ++i) maxerr = std::max(maxerr, (double)fabsf(c[i] - ref[i]));
    std::vector<unsigned long long> st((size_t)grid * 4);
    CK(hipMemcpy(st.data(), g.stamps, st.size() * 8, hipMemcpyDeviceToHost));
    double avg[4] = {0, 0, 0, 0}; unsigned long long mx[4] = {0, 0, 0, 0};
    for (int b = 0; b < nwg; ++b) for (int k = 0; k < 4; ++k) { avg[k] += (double)st[b * 4 + k] / nwg; mx[k] = std::max(mx[k], st[b * 4 + k]); }
    // s_memtime counts at 100 MHz on gfx9-family parts (constant clock): report ticks
    printf("%-34s grid %4d: %7.2f us  %6.1f TF  maxerr %.2e | ticks avg(max): first-stage %5.0f(%llu) loop1 %5.0f(%llu) loop2 %5.0f(%llu) epi %4.0f(%llu)\n",
           label, grid, 1e3 * ms / reps, 2.0 * g.M * g.N * g.K * PASSES / (ms / reps * 1e-3) / 1e12, maxerr,
           avg[0], mx[0], avg[1], mx[1], avg[2], mx[2], avg[3], mx[3]);
}

int main(int argc, char **argv) {
    const int M = 1010, N = 512, K = 2048;
    std::vector<unsigned short> a((size_t)M * K), b((size_t)N * K);
    srand(1);
    for (auto &v : a) v = f2bf((float)rand() / RAND_MAX);
    for (auto &v : b) v = f2bf(((float)rand() / RAND_MAX - 0.5f) * 0.1f);
    std::vector<float> ref((size_t)M * N);
    {
        std::vector<float> af(a.size()), bfv(b.size());
        for (size_t i = 0; i < a.size(); ++i) af[i] = bf2f(a[i]);
        for (size_t i = 0; i < b.size(); ++i) bfv[i] = bf2f(b[i]);
        for (int m = 0; m < M; ++m)
            for (int n = 0; n < N; ++n) {
                double s = 0;
                const float *x = &af[(size_t)m * K], *w = &bfv[(size_t)n * K];
                for (int k = 0; k < K; ++k) s += (double)x[k] * w[k];
                ref[(size_t)m * N + n] = (float)s;
            }
    }
    Gemm g;
    const int pad = argc > 2 ? atoi(argv[2]) : 0;          // extra elements per row
    const int ld = K + pad;
    std::vector<unsigned short> ap((size_t)M * ld, 0), bp((size_t)N * ld, 0);
    for (int m = 0; m < M; ++m) memcpy(&ap[(size_t)m * ld], &a[(size_t)m * K], K * 2);
    for (int n = 0; n < N; ++n) memcpy(&bp[(size_t)n * ld], &b[(size_t)n * K], K * 2);
    unsigned short *dA, *dB, *dZ; float *dC; unsigned long long *dS;
    CK(hipMalloc(&dA, ap.size() * 2)); CK(hipMalloc(&dB, bp.size() * 2)); CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMalloc(&dS, 4096 * 4 * 8)); CK(hipMalloc(&dZ, 4096)); CK(hipMemset(dZ, 0, 4096));
    CK(hipMemcpy(dA, ap.data(), ap.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, bp.data(), bp.size() * 2, hipMemcpyHostToDevice));
    g.A = dA; g.B = dB; g.C = dC; g.M = M; g.N = N; g.K = K; g.ld = ld; g.stamps = dS; g.zeros = dZ;
    printf("row stride %d elements (%d bytes)\n", ld, ld * 2);
    const int R = 50;
    setvbuf(stdout, nullptr, _IOLBF, 0);
    const int which = argc > 1 ? atoi(argv[1]) : -1;
    int idx = 0;
#define RUN(label, ...) do { if (which < 0 || which == idx) run<__VA_ARGS__>(label, g, ref, R); ++idx; } while (0)
    RUN("32x64 wk4 nbuf2", 1, 2, 4, 2, 0, 1);
    RUN("32x64 wk4 nbuf3", 1, 2, 4, 3, 0, 1);
    RUN("32x64 wk4 nbuf4", 1, 2, 4, 4, 0, 1);
    RUN("32x64 wk4 nbuf6", 1, 2, 4, 6, 0, 1);
    RUN("32x64 wk4 nbuf3 touch-prefetch", 1, 2, 4, 3, 1, 1);
    RUN("32x64 wk4 nbuf4 touch-prefetch", 1, 2, 4, 4, 1, 1);
    RUN("32x64 wk4 nbuf3 two passes", 1, 2, 4, 3, 0, 2);
    RUN("32x64 wk4 nbuf4 two passes", 1, 2, 4, 4, 0, 2);
    RUN("32x64 wk2 (4 waves) nbuf3", 1, 2, 2, 3, 0, 1);
    RUN("32x64 wk2 (4 waves) nbuf3 touch", 1, 2, 2, 3, 1, 1);
    RUN("64x64 wk2 nbuf3 (128 wg)", 2, 2, 2, 3, 0, 1);
    RUN("64x64 wk2 nbuf3 touch (128 wg)", 2, 2, 2, 3, 1, 1);
    RUN("32x32 wk4 nbuf3 (512 wg)", 1, 1, 4, 3, 0, 1);
    RUN("32x32 wk4 nbuf3 touch (512 wg)", 1, 1, 4, 3, 1, 1);
    RUN("32x32 wk8 nbuf3 (512 wg)", 1, 1, 8, 3, 0, 1);
    return 0;
}
