// Standalone microbenchmark (not part of the library): how fast can ONE CU fill its LDS from global memory on gfx950, by loader?
//   MODE 0  LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B = one 1 KiB piece per wave-instruction), what ta3n::gemm_tiles uses
//   MODE 1  register-staged: global_load_dwordx4 into VGPRs, ds_write_b128 into LDS (what the vendor GEMM kernels do)
// by waves per workgroup (4 / 8 / 16), by pieces each wave keeps in flight (DEPTH), by workgroups per CU (grid 256 or 512), and by
// where the bytes come from:
//   hot     every workgroup re-streams the same 1 MiB (L2-resident after the first pass)
//   cold    every workgroup streams its own region, read once (HBM / fabric)
//   shared  the four workgroups b, b+8, b+16, b+24 - same XCD - stream the same region at the same time (a GEMM's column tiles sharing
//           a row panel: one fabric read, three L2 hits-on-miss)
// No compute, no barriers: each wave fills its own LDS slice.  Prints GB/s per CU and B/clk/CU (at the 2.4 GHz the chip sustains).
// Round 4: the production K loops see 10.6 / 14.7 / 21 B/clk/CU from 4 / 8 / 2x8 DMA-issuing waves; this isolates the
// loader from everything else in those kernels.
// build: hipcc -O3 --offload-arch=gfx950 tools/proto_fill.hip -o gpurun_out/proto_fill ; run: gpurun_out/proto_fill
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lptr_t;

__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_byte_addr) {
    unsigned keep;
    lds_byte_addr = __builtin_amdgcn_readfirstlane(lds_byte_addr);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// region_of(b): which region workgroup b streams; bytes: per workgroup (a multiple of NW KiB)
template <int NW, int MODE, int DEPTH>
__global__ __launch_bounds__(64 * NW) void fill(const unsigned char *__restrict__ src, size_t region_bytes, int pattern, int n_regions,
                                                 size_t bytes, float *sink) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[DEPTH * NW * 1024];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lptr_t *)lds);
    const int b = blockIdx.x;
    int region = 0;
    if (pattern == 1) region = b % n_regions;
    else if (pattern == 2) region = ((b / 32) * 8 + (b % 8)) % n_regions;
    const unsigned char *base = src + (size_t)region * region_bytes;
    const int rounds = (int)(bytes / (NW * 1024));              // each round: every wave moves one 1 KiB piece
    const size_t wrap = region_bytes;                            // (hot pattern: bytes > region_bytes, wrap around)
    if constexpr (MODE == 0) {
#pragma unroll 1
        for (int r = 0; r < rounds; ++r) {
            const size_t off = (((size_t)r * NW + wave) * 1024) % wrap + lane * 16;
            glds16(base + off, lds_base + (unsigned)(((r % DEPTH) * NW + wave) * 1024));
            wait_vm<DEPTH - 1>();                                // at most DEPTH pieces of this wave in flight
        }
        wait_vm<0>();
    } else {
        u32x4 buf[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const size_t off = (((size_t)d * NW + wave) * 1024) % wrap + lane * 16;
            buf[d] = *reinterpret_cast<const u32x4 *>(base + off);
        }
#pragma unroll 1
        for (int r = 0; r + DEPTH <= rounds; r += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                *reinterpret_cast<u32x4 *>(lds + ((d * NW + wave) * 1024) + lane * 16) = buf[d];      // (the compiler counts vmcnt down to this piece)
                const size_t off = (((size_t)(r + DEPTH + d) * NW + wave) * 1024) % wrap + lane * 16;
                buf[d] = *reinterpret_cast<const u32x4 *>(base + off);
            }
        }
    }
    __syncthreads();
    if (tid == 0) sink[b] = (float)lds[(b * 16) % (DEPTH * NW * 1024)];
}

struct Bufs { unsigned char *src; float *sink; size_t total; };

template <int NW, int MODE, int DEPTH>
static void run(const Bufs &B, int pattern, int grid) {
    const size_t region_bytes = pattern == 0 ? (size_t)1 << 20 : (size_t)4 << 20;      // hot: 1 MiB shared by all; else 4 MiB regions
    const int n_regions = pattern == 0 ? 1 : (int)(B.total / region_bytes) - 1;       // (- 1: the register path reads DEPTH pieces past the end)
    const size_t bytes = pattern == 0 ? (size_t)4 << 20 : (size_t)2 << 20;             // per workgroup
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        // cold patterns: odd repetitions read the regions' second halves, even ones the first halves again - by then >= 512 MiB of other
        // bytes have gone through the 256 MiB last-level cache
        const unsigned char *src = B.src + (pattern == 0 ? 0 : (size_t)(rep & 1) * (region_bytes / 2));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((fill<NW, MODE, DEPTH>), dim3(grid), dim3(64 * NW), 0, 0, src, region_bytes, pattern, n_regions - 4, bytes, B.sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 || pattern != 0) best = ms < best ? ms : best;      // (hot: the first pass warms the L2s)
    }
    const double total = (double)bytes * grid, s = best * 1e-3;
    const double per_cu = total / s / 256.0;
    printf("%-6s %-4s waves %2d depth %d grid %3d : %7.1f us  %6.2f TB/s chip  %6.1f GB/s per CU  %5.1f B/clk/CU\n",
           pattern == 0 ? "hot" : pattern == 1 ? "cold" : "shared", MODE == 0 ? "dma" : "reg", NW, DEPTH, grid, best * 1e3, total / s / 1e12,
           per_cu / 1e9, per_cu / 2.4e9);
    fflush(stdout);
}

template <int NW, int MODE>
static void sweep_depth(const Bufs &B, int pattern, int grid) {
    run<NW, MODE, 2>(B, pattern, grid);
    run<NW, MODE, 4>(B, pattern, grid);
    run<NW, MODE, 8>(B, pattern, grid);
}

// `proto_fill middle` (round 6): the weight-streaming floor of a FUSED relation-level middle (Hr GEMM -> heads -> gR
// dgrad as one kernel per block of videos).  Such a workgroup owns whole rows, so it must pull EVERY weight matrix of the middle through
// its own LDS: 4 x W1_j (fwd) + W_dv + 4 x W1_j (bwd) = 9 x 128 KiB of bf16 = 1.125 MiB, all workgroups the same bytes (L2 / MALL
// resident).  With 202 videos there are 13 / 26 / 52 / 104 such workgroups (16 / 8 / 4 / 2 videos each) on 256 CUs.  No compute.
template <int NW, int MODE>
static void run_middle(const Bufs &B, int grid) {
    const size_t region_bytes = (size_t)1152 << 10, bytes = region_bytes;      // every workgroup streams the same 1.125 MiB once
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((fill<NW, MODE, 4>), dim3(grid), dim3(64 * NW), 0, 0, B.src, region_bytes, 0, 1, bytes, B.sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0) best = ms < best ? ms : best;
    }
    printf("middle %-4s waves %2d grid %3d (%2d videos per workgroup): %6.1f us to stream 1.125 MiB of weights per workgroup = %5.1f GB/s = %4.1f B/clk per workgroup\n",
           MODE == 0 ? "dma" : "reg", NW, grid, (202 + grid - 1) / grid, best * 1e3, bytes / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 2.4e9);
    fflush(stdout);
}

int main(int argc, char **argv) {
    if (argc > 1 && argv[1][0] == 'm') {
        Bufs B;
        B.total = (size_t)8 << 20;
        CK(hipMalloc(&B.src, B.total));
        CK(hipMalloc(&B.sink, 4096 * sizeof(float)));
        CK(hipMemset(B.src, 1, B.total));
        CK(hipDeviceSynchronize());
        for (int grid : {13, 26, 52, 104, 202}) {
            run_middle<8, 0>(B, grid); run_middle<16, 0>(B, grid); run_middle<8, 1>(B, grid); run_middle<16, 1>(B, grid);
        }
        return 0;
    }
    Bufs B;
    B.total = (size_t)3 << 30;      // 3 GiB: 768 regions of 4 MiB - every cold workgroup of a 512-workgroup grid has its own
    CK(hipMalloc(&B.src, B.total));
    CK(hipMalloc(&B.sink, 4096 * sizeof(float)));
    CK(hipMemset(B.src, 1, B.total));
    CK(hipDeviceSynchronize());
    for (int pattern = 0; pattern < 3; ++pattern)
        for (int grid = 256; grid <= 512; grid += 256) {
            sweep_depth<4, 0>(B, pattern, grid);
            sweep_depth<4, 1>(B, pattern, grid);
            sweep_depth<8, 0>(B, pattern, grid);
            sweep_depth<8, 1>(B, pattern, grid);
            if (grid == 256) { sweep_depth<16, 0>(B, pattern, grid); sweep_depth<16, 1>(B, pattern, grid); }
        }
    return 0;
}
