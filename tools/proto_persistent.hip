// Standalone microbenchmark (not part of the library): what would a PERSISTENT step kernel pay between dependency levels?
// One launch of G workgroups x 256 threads runs L "levels"; between two levels every workgroup passes a grid barrier
// (agent-scope counter in device memory: one relaxed add per workgroup, wave 0 polls, then an acquire).  Variants:
//   work = 0   barrier only (the control path: atomic round trip + poll)
//   work = W   every workgroup first writes W KiB of fresh output with ordinary stores and reads W KiB that ANOTHER workgroup
//              wrote in the previous level - so the release has dirty lines to write back from this XCD's L2 and the acquire
//              really invalidates (what a level of the train step does: 4-16 KiB tiles)
//   wt  = 1    the output is written with sc0 sc1 write-through stores instead (no dirty L2 lines at the release)
// Reference point on the same box: the same levels as separate kernel launches on one stream (kernel boundary = implicit
// release / acquire + dispatch).
// build: hipcc -O3 --offload-arch=gfx950 tools/proto_persistent.hip -o gpurun_out/proto_persistent
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void st_wt(float *p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }

// one level's work of one workgroup: read `kib` KiB written by workgroup (b + 37) % G in the previous level, write `kib` KiB
template <bool WT>
__device__ __forceinline__ float level_work(float *buf, int level, int G, int kib, float carry) {
    const int tid = threadIdx.x, b = blockIdx.x;
    const size_t per_wg = (size_t)kib * 256;                 // floats
    const float *src = buf + ((size_t)((level + 1) & 1) * G + (b + 37) % G) * per_wg;
    float *dst = buf + ((size_t)(level & 1) * G + b) * per_wg;
    f32x4 acc = {carry, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)tid * 4; i < per_wg; i += 1024) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(src + i);
        acc += v;
    }
    for (size_t i = (size_t)tid * 4; i < per_wg; i += 1024) {
        const f32x4 v = {acc[0] + (float)i, acc[1], acc[2], acc[3]};
        if (WT) st_wt(dst + i, v); else *reinterpret_cast<f32x4 *>(dst + i) = v;
    }
    return acc[0] + acc[1] + acc[2] + acc[3];
}

template <bool WT, bool TREE>
__global__ __launch_bounds__(256) void persistent(float *buf, int *counter, int levels, int G, int kib, int *err) {
    float carry = 0.f;
    for (int l = 0; l < levels; ++l) {
        if (kib) carry = level_work<WT>(buf, l, G, kib, carry) * 1e-30f;
        // ---- grid barrier ----
        // all stores of the workgroup have left the CU (vmcnt(0)); ONE thread then makes them agent-visible (release: write back
        // this XCD's dirty L2 lines - nothing to do for write-through stores), bumps the arrival counter; the last arriver
        // publishes the epoch in a separate line that the others poll with a back-off (polling the counter itself delays the adds)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            if (!WT) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            bool last;
            if (TREE) {        // two levels: 16 group counters (their own 128-byte lines), the last arriver of a group bumps the root
                const int grp = blockIdx.x & 15, gsize = G / 16;
                last = __hip_atomic_fetch_add(counter + 64 + 32 * grp, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize * (l + 1) - 1;
                if (last) last = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 16 * (l + 1) - 1;
            } else {
                last = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == G * (l + 1) - 1;
            }
            if (last) {
                __hip_atomic_store(counter + 32, l + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                int spins = 0;
                while (__hip_atomic_load(counter + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < l + 1) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > (1 << 22)) { *err = 1; break; }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
    if (carry == 123.f) buf[0] = carry;
}

template <bool WT>
__global__ __launch_bounds__(256) void one_level(float *buf, int level, int G, int kib) {
    float carry = 0.f;
    if (kib) carry = level_work<WT>(buf, level, G, kib, carry);
    if (carry == 123.f) buf[0] = carry;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("%s, %d CUs\n", prop.gcnArchName, prop.multiProcessorCount);
    const int L = 64, REP = 20;
    float *buf; int *counter, *err;
    CK(hipMalloc(&buf, (size_t)2 * 512 * 64 * 1024));
    CK(hipMemset(buf, 0, (size_t)2 * 512 * 64 * 1024));
    CK(hipMalloc(&counter, 4096)); CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("%-10s %-6s %-4s %-22s %-22s %-26s\n", "workgroups", "KiB/wg", "wt", "persistent us/level", "... tree barrier", "separate launches us/level");
    for (int G : {256, 512}) {
        for (int kib : {0, 4, 16, 64}) {
            for (int wt = 0; wt < 2; ++wt) {
                if (kib == 0 && wt) continue;
                float ms_p = 0.f, ms_s = 0.f, ms_t = 0.f;
                for (int r = 0; r < REP + 2; ++r) {             // 2 warm-up repetitions
                    CK(hipMemsetAsync(counter, 0, 4096, 0));
                    CK(hipEventRecord(e0, 0));
                    if (wt) persistent<true, false><<<G, 256>>>(buf, counter, L, G, kib, err);
                    else persistent<false, false><<<G, 256>>>(buf, counter, L, G, kib, err);
                    CK(hipEventRecord(e1, 0));
                    CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (r >= 2) ms_p += ms;
                    CK(hipMemsetAsync(counter, 0, 4096, 0));
                    CK(hipEventRecord(e0, 0));
                    if (wt) persistent<true, true><<<G, 256>>>(buf, counter, L, G, kib, err);
                    else persistent<false, true><<<G, 256>>>(buf, counter, L, G, kib, err);
                    CK(hipEventRecord(e1, 0));
                    CK(hipEventSynchronize(e1));
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (r >= 2) ms_t += ms;
                    CK(hipEventRecord(e0, 0));
                    for (int l = 0; l < L; ++l) {
                        if (wt) one_level<true><<<G, 256>>>(buf, l, G, kib);
                        else one_level<false><<<G, 256>>>(buf, l, G, kib);
                    }
                    CK(hipEventRecord(e1, 0));
                    CK(hipEventSynchronize(e1));
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (r >= 2) ms_s += ms;
                }
                int h_err = 0; CK(hipMemcpy(&h_err, err, 4, hipMemcpyDeviceToHost));
                printf("%-10d %-6d %-4d %-22.2f %-22.2f %-26.2f%s\n", G, kib, wt, ms_p * 1e3 / (REP * L), ms_t * 1e3 / (REP * L), ms_s * 1e3 / (REP * L), h_err ? "  (barrier timed out!)" : "");
            }
        }
    }
    return 0;
}
