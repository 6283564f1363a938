// Launcher of the tile-list GEMM (the kernel template lives in ta3n_gemm_kernel.h, its instantiations in ta3n_gemm_i*.hip).
#include "ta3n_gemm_kernel.h"
#include "ta3n_plan.h"      // set_error

namespace ta3n {
// every instantiation is defined in one of ta3n_gemm_i*.hip
#define TA3N_EXTERN(wm, wn, wk) \
    extern template __global__ void gemm_tiles<wm, wn, wk, 0, 2, 1, 1>(const Task *, const Seg *, Ptrs, int, int, int, SgdSide, const Wait *, int, int, int, int); \
    extern template __global__ void gemm_tiles<wm, wn, wk, 1, 2, 1, 1>(const Task *, const Seg *, Ptrs, int, int, int, SgdSide, const Wait *, int, int, int, int); \
    extern template __global__ void gemm_tiles<wm, wn, wk, 1, 3, 1, 1>(const Task *, const Seg *, Ptrs, int, int, int, SgdSide, const Wait *, int, int, int, int); \
    extern template __global__ void gemm_tiles<wm, wn, wk, 2, 2, 1, 1>(const Task *, const Seg *, Ptrs, int, int, int, SgdSide, const Wait *, int, int, int, int); \
    extern template __global__ void gemm_tiles<wm, wn, wk, 2, 3, 1, 1>(const Task *, const Seg *, Ptrs, int, int, int, SgdSide, const Wait *, int, int, int, int); \
    extern template __global__ void gemm_tiles<wm, wn, wk, 3, 2, 1, 1>(const Task *, const Seg *, Ptrs, int, int, int, SgdSide, const Wait *, int, int, int, int); \
    extern template __global__ void gemm_tiles<wm, wn, wk, 3, 3, 1, 1>(const Task *, const Seg *, Ptrs, int, int, int, SgdSide, const Wait *, int, int, int, int); \
    extern template __global__ void gemm_tiles<wm, wn, wk, 4, 2, 1, 1>(const Task *, const Seg *, Ptrs, int, int, int, SgdSide, const Wait *, int, int, int, int); \
    extern template __global__ void gemm_tiles<wm, wn, wk, 4, 3, 1, 1>(const Task *, const Seg *, Ptrs, int, int, int, SgdSide, const Wait *, int, int, int, int);
TA3N_TILE_CONFIGS(TA3N_EXTERN)
#define TA3N_EXTERN_BLOCKED(wm, wn, wk, rm, rn, ns) \
    extern template __global__ void gemm_tiles<wm, wn, wk, 2, ns, rm, rn>(const Task *, const Seg *, Ptrs, int, int, int, SgdSide, const Wait *, int, int, int, int);
TA3N_BLOCKED_CONFIGS(TA3N_EXTERN_BLOCKED)
#define TA3N_EXTERN_KIND(wm, wn, wk, bf, ns, kv) \
    extern template __global__ void gemm_tiles<wm, wn, wk, bf, ns, 1, 1, kv>(const Task *, const Seg *, Ptrs, int, int, int, SgdSide, const Wait *, int, int, int, int);
TA3N_KIND_CONFIGS(TA3N_EXTERN_KIND)
#define TA3N_EXTERN_HS(wm, wn, wk, rm, rn, ns) \
    extern template __global__ void gemm_tiles<wm, wn, wk, 5, ns, rm, rn>(const Task *, const Seg *, Ptrs, int, int, int, SgdSide, const Wait *, int, int, int, int);
TA3N_HS_CONFIGS(TA3N_EXTERN_HS)

// (-DTA3N_GEMM_STAMPS: the stamps live in the workspace region "stamps" - ta3n_gemm_kernel.h; tools/gemm_stamps.py reads them there)

bool tile_config_ok(int cfg) {
    // optional ten-thousands digit: register blocking of the bf16-twin kernel (1: 2 row blocks per wave, 2: 2 column blocks, 3: 2 x 2)
    const int blk = cfg / 10000;
    cfg %= 10000;
    const int stages = cfg / 1000;   // optional thousands digit: LDS stages of the bf16 kernel (0 = plan's choice); 5 / 6 / 7: HALF stages (64 k), 2 / 3 / 4 of them
    cfg %= 1000;
    if (stages >= 5 && stages <= 7) {
        if (blk < 0 || blk > 5) return false;
        const int rm = blk_rm(blk), rn = blk_rn(blk);
#define TA3N_CHECK_HS(wm, wn, wk, rm_, rn_, ns) \
        if (cfg == wm * 100 + wn * 10 + wk && rm == rm_ && rn == rn_ && stages - 3 == ns) return true;
        TA3N_HS_CONFIGS(TA3N_CHECK_HS)
        return false;
    }
    if (stages != 0 && stages != 2 && stages != 3) return false;
    if (blk != 0) {
        if (blk < 0 || blk > 3) return false;
        const int rm = 1 + (blk & 1), rn = 1 + (blk >> 1);
#define TA3N_CHECK_BLOCKED(wm, wn, wk, rm_, rn_, ns) \
        if (cfg == wm * 100 + wn * 10 + wk && rm == rm_ && rn == rn_ && (stages == 0 || stages == ns)) return true;
        TA3N_BLOCKED_CONFIGS(TA3N_CHECK_BLOCKED)
        return false;
    }
#define TA3N_CHECK(wm, wn, wk) if (cfg == wm * 100 + wn * 10 + wk) return true;
    TA3N_TILE_CONFIGS(TA3N_CHECK)
    return false;
}

int launch_gemm(const Phase &ph, const Task *d_tasks, const Seg *d_segs, const Ptrs &ptrs, int hyper_off,
                int zeros_off, int twin_off, hipStream_t stream, const SgdSide *side, const Wait *d_waits, int pair_delta, int kinds) {
    const int chain_off = d_waits ? ph.chain_off : -1;
    if (ph.chain_off >= 0 && !d_waits) return -4;
    // measurement knobs of the hand-off protocol (defaults = the shipped protocol): TA3N_CHAIN_SLEEP = poll back-off in units of 512
    // cycles; TA3N_CHAIN_NOACQ = 1: no acquire fence after the poll; TA3N_CHAIN_MEMSET = 1: counters reset by a memset before the launch
    static const int knobs_env = [] {
        const char *a = getenv("TA3N_CHAIN_SLEEP"), *b = getenv("TA3N_CHAIN_NOACQ"), *c = getenv("TA3N_CHAIN_MEMSET");
        const char *d = getenv("TA3N_CHAIN_NOWAIT"), *e = getenv("TA3N_CHAIN_RMWPOLL");
        return ((a ? atoi(a) : 1) & 255) | ((b && atoi(b)) ? 256 : 0) | ((c && atoi(c)) ? 512 : 0) | ((d && atoi(d)) ? 1024 : 0) |
               ((e && atoi(e)) ? 2048 : 0);
    }();
    const int knobs = knobs_env;
    if (chain_off >= 0 && (knobs & 512)) {
        if (hipMemsetAsync(ptrs.ws + chain_off, 0, sizeof(int) * (size_t)(2 + ph.chain_n), stream) != hipSuccess) return -2;
    }
    if (ph.task_count == 0) return 0;
    // chained launches, split-K tiles and the optimiser inside the gradient tiles were built, measured and rejected; their
    // device code lives in the experiments build only (ta3n_kernels.h: TA3N_EXPERIMENTS).  The plan BUILDER still emits such launch lists
    // (host code, checked on the CPU by tests/plan_interp.py); launching one on the default library is an error, not a silent fallback.
    if (!TA3N_EXPERIMENTS && (chain_off >= 0 || (kinds & 32) || (side && side->p_new != nullptr))) {
        set_error("this launch list uses ta3n_config.chain / split_k or the fused update: build with -DTA3N_EXPERIMENTS=1 "
                  "(TA3N_LIBDIR=ta3n_amd/lib_ab TA3N_EXTRA_FLAGS=-DTA3N_EXPERIMENTS=1 python -m ta3n_amd.build)");
        return -5;
    }
    SgdSide sd;
    std::memset(&sd, 0, sizeof(sd));
    sd.p16_off = -1;
    if (side) sd = *side;
    const int chain_n = ph.chain_n;
    const dim3 grid(ph.task_count);
    const Task *tp = d_tasks + ph.task_begin;
    const int cfg = ph.wm * 100 + ph.wn * 10 + ph.wk;
    bool launched = false;
    const int rm = ph.rm > 0 ? ph.rm : 1, rn = ph.rn > 0 ? ph.rn : 1;
    if (ph.bf16 & 64) {        // half stages (the plan only sets the bit on launches that read plain bf16 twins)
        if ((ph.bf16 & 48) != 16) return -3;
#define TA3N_LAUNCH_HS(wm, wn, wk, rm_, rn_, ns)                                                                           \
        if (!launched && cfg == wm * 100 + wn * 10 + wk && rm == rm_ && rn == rn_ && (ph.bf16 & 15) == ns) {               \
            hipLaunchKernelGGL((gemm_tiles<wm, wn, wk, 5, ns, rm_, rn_>), grid, dim3(64 * wm * wn * wk), 0, stream, tp, d_segs, \
                               ptrs, hyper_off, zeros_off, twin_off, sd, d_waits, chain_off, chain_n, knobs, pair_delta);   \
            launched = true;                                                                                               \
        }
        TA3N_HS_CONFIGS(TA3N_LAUNCH_HS)
        if (!launched) return -1;
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    if (rm * rn > 1) {
        if (!(ph.bf16 & 16)) return -3;     // (the plan builder never emits this: blocked tiles read bf16 twins)
#define TA3N_LAUNCH_BLOCKED(wm, wn, wk, rm_, rn_, ns)                                                                      \
        if (!launched && cfg == wm * 100 + wn * 10 + wk && rm == rm_ && rn == rn_ && (ph.bf16 & 15) == ns) {               \
            hipLaunchKernelGGL((gemm_tiles<wm, wn, wk, 2, ns, rm_, rn_>), grid, dim3(64 * wm * wn * wk), 0, stream, tp, d_segs, \
                               ptrs, hyper_off, zeros_off, twin_off, sd, d_waits, chain_off, chain_n, knobs, pair_delta);                                      \
            launched = true;                                                                                               \
        }
        TA3N_BLOCKED_CONFIGS(TA3N_LAUNCH_BLOCKED)
        if (!launched) return -1;
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    // a kernel that holds only the K-loop variants this launch's tasks use, where one is built (kinds: bit mask, 0 = unknown)
    static const bool kind_kernels = [] { const char *e = getenv("TA3N_KIND_KERNELS"); return !(e && atoi(e) == 0); }();
    if (kind_kernels && kinds != 0 && !(kinds & 32) && rm * rn == 1 && chain_off < 0 && sd.p_new == nullptr) {
        // gemm_tiles' MODE and stage count as the plain dispatch below picks them
        const int mode = ph.bf16 == 0 ? 0 : (ph.bf16 & 16) ? ((ph.bf16 & 32) ? 4 : 2) : (ph.bf16 & 32) ? 3 : 1;
        const int ns = ph.bf16 == 0 ? 2 : ((ph.bf16 & 15) == 3 ? 3 : 2);
#define TA3N_LAUNCH_KIND(wm, wn, wk, bf, ns_, kv)                                                                              \
        if (!launched && cfg == wm * 100 + wn * 10 + wk && mode == bf && ns == ns_ && (kinds & ~kv) == 0) {                    \
            hipLaunchKernelGGL((gemm_tiles<wm, wn, wk, bf, ns_, 1, 1, kv>), grid, dim3(64 * wm * wn * wk), 0, stream, tp, d_segs, \
                               ptrs, hyper_off, zeros_off, twin_off, sd, d_waits, chain_off, chain_n, knobs, pair_delta);          \
            launched = true;                                                                                                   \
        }
        TA3N_KIND_CONFIGS(TA3N_LAUNCH_KIND)
        if (launched) return hipGetLastError() == hipSuccess ? 0 : -2;
    }
#define TA3N_LAUNCH_ONE(wm, wn, wk, bf, ns)                                                                         \
    hipLaunchKernelGGL((gemm_tiles<wm, wn, wk, bf, ns, 1, 1>), grid, dim3(64 * wm * wn * wk), 0, stream, tp, d_segs, ptrs, \
                       hyper_off, zeros_off, twin_off, sd, d_waits, chain_off, chain_n, knobs, pair_delta)
#define TA3N_LAUNCH(wm, wn, wk)                                   \
    if (cfg == wm * 100 + wn * 10 + wk) {                         \
        switch (ph.bf16) {                                        \
            case 0: TA3N_LAUNCH_ONE(wm, wn, wk, 0, 2); break;     \
            case 3: TA3N_LAUNCH_ONE(wm, wn, wk, 1, 3); break;     \
            case 18: TA3N_LAUNCH_ONE(wm, wn, wk, 2, 2); break;    \
            case 19: TA3N_LAUNCH_ONE(wm, wn, wk, 2, 3); break;    \
            case 34: TA3N_LAUNCH_ONE(wm, wn, wk, 3, 2); break;    \
            case 35: TA3N_LAUNCH_ONE(wm, wn, wk, 3, 3); break;    \
            case 50: TA3N_LAUNCH_ONE(wm, wn, wk, 4, 2); break;    \
            case 51: TA3N_LAUNCH_ONE(wm, wn, wk, 4, 3); break;    \
            default: TA3N_LAUNCH_ONE(wm, wn, wk, 1, 2); break;    \
        }                                                         \
        launched = true;                                          \
    }
    TA3N_TILE_CONFIGS(TA3N_LAUNCH)
    if (!launched) return -1;
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace ta3n
