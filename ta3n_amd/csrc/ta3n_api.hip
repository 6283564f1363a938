// C ABI of libta3n_hip.so (see include/ta3n_hip.h): plan lifetime, layout
// queries and the per-step launchers.  Launchers only enqueue work on the
// caller's stream; they never synchronise, so a whole train step can be captured
// into a hipGraph by the host (the reference's shapes are static thanks to its
// own pad-to-batch-size rule, main.py:359-364).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/ta3n_hip.h"
#include "ta3n_kernels.h"
#include "ta3n_plan.h"

using namespace ta3n;

static_assert(sizeof(ta3n_hyper) == sizeof(ta3n::Hyper), "public and device hyper structs must match");
static_assert(sizeof(ta3n_hyper) <= 32 * sizeof(float), "hyper region is 32 floats");

namespace {
thread_local std::string g_err;

int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail(TA3N_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

int ensure_uploaded(ta3n_plan *p) {
    if (p->uploaded) return TA3N_OK;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    (void)st;
    HIP_TRY(hipGetDevice(&p->device));
    HIP_TRY(hipMalloc(&p->d_segs, p->segs.size() * sizeof(Seg)));
    HIP_TRY(hipMalloc(&p->d_tasks, p->tasks.size() * sizeof(Task)));
    HIP_TRY(hipMemcpy(p->d_segs, p->segs.data(), p->segs.size() * sizeof(Seg), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(p->d_tasks, p->tasks.data(), p->tasks.size() * sizeof(Task), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc(&p->d_waits, (p->waits.size() + 1) * sizeof(Wait)));      // (+ 1: never a zero-byte allocation)
    if (!p->waits.empty()) HIP_TRY(hipMemcpy(p->d_waits, p->waits.data(), p->waits.size() * sizeof(Wait), hipMemcpyHostToDevice));
    p->phase_kinds.assign(p->phases.size(), 0);
    for (size_t i = 0; i < p->phases.size(); ++i) {
        const Phase &ph = p->phases[i];
        if (ph.kind != PH_GEMM) continue;
        int m = 0;
        for (int k = ph.task_begin; k < ph.task_begin + ph.task_count; ++k) {
            const Task &t = p->tasks[k];
            if (t.epi & EPI_SPLITK) m |= 32;      // (optional epilogue paths: only the full kernels hold them)
            if (t.seg_count == 0 || (t.epi & (EPI_SGD | EPI_COLSUM))) continue;
            const int kind = t.seg0.a_kmajor * 2 + t.seg0.b_kmajor;
            m |= kind < 3 ? (1 << kind) : ((t.epi & EPI_ROWSUM_A) ? 16 : 8);
        }
        p->phase_kinds[i] = m;
    }
    p->uploaded = true;
    return TA3N_OK;
}

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// buffers of one launch sequence; the parameter twins it reads are those of region `twin_region` (0: "p16", 1: "p16b")
Ptrs make_ptrs(const ta3n_plan *p, const float *x, const float *params, float *grads, float *ws, int twin_region = 0) {
    const int32_t o = twin_region ? p->geom.o_p16b : p->geom.o_p16;
    return Ptrs{x, params, grads, ws, (ws && o >= 0) ? ws + o : nullptr};
}

int run_group(ta3n_plan *p, int group, const Ptrs &ptrs, float *params_rw, float *momentum, hipStream_t stream,
              hipEvent_t join_after_first = nullptr, int first_launch = 0, int n_launches = 1 << 30,
              const SgdSide *side = nullptr) {
    bool first = true;
    int index = -1;
    for (const Phase &ph : p->phases) {
        if (ph.group != group) continue;
        const int kinds = p->phase_kinds.empty() ? 0 : p->phase_kinds[&ph - p->phases.data()];
        ++index;
        if (index < first_launch || index >= first_launch + n_launches) continue;
        if (!first && join_after_first) {   // everything after the first launch also depends on work the caller put on another stream
            if (hipStreamWaitEvent(stream, join_after_first, 0) != hipSuccess) return fail(TA3N_ERR_HIP, "hipStreamWaitEvent failed");
            join_after_first = nullptr;
        }
        first = false;
        int rc = 0;
        switch (ph.kind) {
            case PH_GEMM:
                rc = launch_gemm(ph, static_cast<const Task *>(p->d_tasks), static_cast<const Seg *>(p->d_segs), ptrs,
                                 p->geom.o_hyper, p->geom.o_zeros, p->geom.o_ws16, stream, side, static_cast<const Wait *>(p->d_waits),
                                 p->geom.pair_delta, kinds);
                break;
            case PH_POOL_FWD: rc = launch_pool_fwd(p->geom, ptrs, stream); break;
            case PH_LOSS: rc = launch_loss(p->geom, ptrs, stream); break;
            case PH_POOL_BWD: rc = launch_pool_bwd(p->geom, ptrs, stream); break;
            case PH_HEADS: rc = launch_heads(p->geom, ptrs, stream); break;
            case PH_POOL_CLS: rc = launch_pool_cls(p->geom, ptrs, stream); break;
            case PH_POOL_AVG_FWD: rc = launch_pool_avg_fwd(p->geom, ptrs, stream); break;
            case PH_POOL_AVG_BWD: rc = launch_pool_avg_bwd(p->geom, ptrs, stream); break;
            case PH_BN_FWD: rc = launch_bn_shared_fwd(p->geom, ptrs, stream); break;
            case PH_BN_BWD: rc = launch_bn_shared_bwd(p->geom, ptrs, stream); break;
            case PH_GRAD_NORM: rc = launch_grad_norm(p->geom, ptrs.g, ptrs.ws, stream); break;
            case PH_SGD: rc = launch_sgd(p->geom, params_rw, ptrs.g, momentum, ptrs.ws, stream); break;
            default: rc = -1;
        }
        if (rc == -5) return TA3N_ERR_INVALID;      // (launch_gemm has set the message: an experiments-only launch list on the default library)
        if (rc != 0) return fail(TA3N_ERR_HIP, "kernel launch failed in phase kind " + std::to_string(ph.kind) + ": " +
                                                   hipGetErrorString(hipGetLastError()));
    }
    return TA3N_OK;
}
}  // namespace

void ta3n::set_error(const std::string &msg) { g_err = msg; }

namespace {
// Multiply-adds x 2 of a GEMM launch as the plan tiled it: the valid rows x columns of every tile times the K of its Segs (what the launch
// computes, not what a tile's padding wastes; optimiser side jobs and column-sum tasks count nothing).  For "TFLOP/s of launch i".
double phase_flops(const ta3n_plan &p, const ta3n::Phase &ph) {
    using namespace ta3n;
    if (ph.kind != PH_GEMM) return 0.0;
    const int BM = 32 * ph.wm * (ph.rm > 0 ? ph.rm : 1), BN = 32 * ph.wn * (ph.rn > 0 ? ph.rn : 1);
    double f = 0.0;
    for (int k = ph.task_begin; k < ph.task_begin + ph.task_count; ++k) {
        const Task &t = p.tasks[k];
        if (t.seg_count == 0 || (t.epi & (EPI_SGD | EPI_COLSUM))) continue;
        const int rows = std::max(0, std::min(BM, t.m_valid - t.m0)), cols = std::max(0, std::min(BN, t.n_valid - t.n0));
        int64_t K = 0;
        for (int sidx = t.seg_begin; sidx < t.seg_begin + t.seg_count; ++sidx) K += p.segs[sidx].klen;
        f += 2.0 * rows * cols * (double)K;
    }
    return f;
}
}  // namespace

extern "C" {

const char *ta3n_last_error(void) { return g_err.c_str(); }
const char *ta3n_version(void) {
    return TA3N_EXPERIMENTS ? "ta3n_hip 0.3 (gfx950, fp32 MFMA 32x32x2 | bf16 MFMA 32x32x16 with fp32 accumulation) +experiments"
                            : "ta3n_hip 0.3 (gfx950, fp32 MFMA 32x32x2 | bf16 MFMA 32x32x16 with fp32 accumulation)";
}

int ta3n_plan_create(const ta3n_config *cfg, ta3n_plan **out) {
    if (!cfg || !out) return fail(TA3N_ERR_INVALID, "null argument");
    ta3n_plan *p = new (std::nothrow) ta3n_plan();
    if (!p) return fail(TA3N_ERR_NOMEM, "out of memory");
    p->cfg = *cfg;
    std::string err;
    const int rc = build_plan(*p, err);
    if (rc != TA3N_OK) {
        delete p;
        return fail(rc, err);
    }
    *out = p;
    return TA3N_OK;
}

void ta3n_plan_destroy(ta3n_plan *p) {
    if (!p) return;
    if (p->uploaded) {
        (void)hipFree(p->d_segs);
        (void)hipFree(p->d_tasks);
        (void)hipFree(p->d_waits);
    }
    delete p;
}

int ta3n_num_params(const ta3n_plan *p) { return p ? (int)p->params.size() : TA3N_ERR_INVALID; }

int ta3n_param_info(const ta3n_plan *p, int i, const char **name, int64_t *offset, int32_t *rows, int32_t *cols,
                    int32_t *live) {
    if (!p || i < 0 || i >= (int)p->params.size()) return fail(TA3N_ERR_INVALID, "param index out of range");
    const ParamInfo &pi = p->params[i];
    if (name) *name = pi.name.c_str();
    if (offset) *offset = pi.off;
    if (rows) *rows = pi.rows;
    if (cols) *cols = pi.cols;
    if (live) *live = pi.live ? 1 : 0;
    return TA3N_OK;
}

int64_t ta3n_param_floats(const ta3n_plan *p) { return p ? p->param_floats : -1; }
int64_t ta3n_live_param_floats(const ta3n_plan *p) { return p ? p->live_floats : -1; }
int64_t ta3n_workspace_floats(const ta3n_plan *p) { return p ? p->ws_floats : -1; }

int64_t ta3n_ws_offset(const ta3n_plan *p, const char *region) {
    if (!p || !region) return -1;
    return p->woff(region);
}
int64_t ta3n_ws_size(const ta3n_plan *p, const char *region) {
    if (!p || !region) return -1;
    for (const auto &r : p->regions)
        if (r.name == region) return r.size;
    return -1;
}

int64_t ta3n_plan_describe(const ta3n_plan *p, char *buf, int64_t cap) {
    if (!p) return -1;
    std::ostringstream o;
    o << "{\"config\":{\"Bs\":" << p->geom.Bs << ",\"Bt\":" << p->geom.Bt << ",\"T\":" << p->geom.T << ",\"D\":" << p->geom.D
      << ",\"F\":" << p->geom.F << ",\"NB\":" << p->geom.NB << ",\"C\":" << p->geom.C << ",\"flags\":" << p->geom.flags
      << ",\"n_tuples\":" << p->n_tuples << "},\"param_floats\":" << p->param_floats << ",\"live_floats\":" << p->live_floats
      << ",\"ws_floats\":" << p->ws_floats << ",\"regions\":{";
    for (size_t i = 0; i < p->regions.size(); ++i)
        o << (i ? "," : "") << "\"" << p->regions[i].name << "\":[" << p->regions[i].off << "," << p->regions[i].size << "]";
    o << "},\"phases\":[";
    for (size_t i = 0; i < p->phases.size(); ++i) {
        const Phase &ph = p->phases[i];
        o << (i ? "," : "") << "{\"kind\":" << ph.kind << ",\"group\":" << ph.group << ",\"task_begin\":" << ph.task_begin
          << ",\"task_count\":" << ph.task_count << ",\"tile\":" << (ph.wm * 100 + ph.wn * 10 + ph.wk + 1000 * ph.bf16)
          << ",\"rm\":" << (ph.rm > 0 ? ph.rm : 1) << ",\"rn\":" << (ph.rn > 0 ? ph.rn : 1)
          << ",\"half_stages\":" << ((ph.bf16 & 64) ? 1 : 0)
          << ",\"flops\":" << phase_flops(*p, ph)
          << ",\"chain_counters\":" << (ph.kind == PH_GEMM && ph.chain_off >= 0 ? ph.chain_n : -1) << "}";
    }
    o << "],\"n_tasks\":" << p->tasks.size() << ",\"n_segs\":" << p->segs.size() << "}";
    const std::string s = o.str();
    if (buf && cap > 0) {
        const int64_t n = std::min<int64_t>(cap - 1, (int64_t)s.size());
        std::memcpy(buf, s.data(), (size_t)n);
        buf[n] = 0;
    }
    return (int64_t)s.size();
}

// Raw descriptor arrays for the CPU plan interpreter in tests/ (host memory owned by the plan).
int ta3n_debug_arrays(const ta3n_plan *p, const void **segs, int64_t *n_segs, const void **tasks, int64_t *n_tasks,
                      const void **phases, int64_t *n_phases, const void **geom, const int32_t **tuples,
                      const int32_t **tuple_first) {
    if (!p) return fail(TA3N_ERR_INVALID, "null plan");
    if (segs) *segs = p->segs.data();
    if (n_segs) *n_segs = (int64_t)p->segs.size();
    if (tasks) *tasks = p->tasks.data();
    if (n_tasks) *n_tasks = (int64_t)p->tasks.size();
    if (phases) *phases = p->phases.data();
    if (n_phases) *n_phases = (int64_t)p->phases.size();
    if (geom) *geom = &p->geom;
    if (tuples) *tuples = p->tuples.data();
    if (tuple_first) *tuple_first = p->tuple_first.data();
    return TA3N_OK;
}

int ta3n_chain_status(ta3n_plan *p, const float *ws, void *stream) {
    if (!p || !ws) return fail(TA3N_ERR_INVALID, "null argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    int bad = 0;
    for (size_t i = 0; i < p->phases.size(); ++i) {
        const Phase &ph = p->phases[i];
        if (ph.kind != PH_GEMM || ph.chain_off < 0) continue;
        int32_t head[2] = {0, 0};
        HIP_TRY(hipMemcpyAsync(head, ws + ph.chain_off, sizeof(head), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (head[1] != 0) {
            bad = 1;
            g_err = "chained launch (phase " + std::to_string(i) + "): workgroup " + std::to_string(head[1] - 1) + " gave up waiting for its producers";
        }
        if (head[0] != 0) {
            bad = 1;
            g_err = "chained launch (phase " + std::to_string(i) + "): " + std::to_string(head[0]) + " workgroups counted and the block was not reset";
        }
    }
    return bad;
}

int ta3n_debug_waits(const ta3n_plan *p, const void **waits, int64_t *n_waits) {
    if (!p) return fail(TA3N_ERR_INVALID, "null plan");
    if (waits) *waits = p->waits.data();
    if (n_waits) *n_waits = (int64_t)p->waits.size();
    return TA3N_OK;
}

int ta3n_debug_struct_sizes(int32_t *seg, int32_t *task, int32_t *phase, int32_t *geom, int32_t *hyper) {
    if (seg) *seg = (int32_t)sizeof(Seg);
    if (task) *task = (int32_t)sizeof(Task);
    if (phase) *phase = (int32_t)sizeof(Phase);
    if (geom) *geom = (int32_t)sizeof(Geom);
    if (hyper) *hyper = (int32_t)sizeof(Hyper);
    return TA3N_OK;
}

int ta3n_num_phases(const ta3n_plan *p, int which) {
    if (!p) return TA3N_ERR_INVALID;
    int n = 0;
    for (const Phase &ph : p->phases)
        if (ph.group == which) ++n;
    return n;
}

int ta3n_init_workspace(ta3n_plan *p, float *ws, void *stream) {
    if (!p || !ws) return fail(TA3N_ERR_INVALID, "null argument");
    if (!aligned16(ws)) return fail(TA3N_ERR_INVALID, "workspace must be 16-byte aligned");
    int rc = ensure_uploaded(p);
    if (rc != TA3N_OK) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(hipMemsetAsync(ws, 0, (size_t)p->ws_floats * sizeof(float), s));
    if (launch_fill(ws + p->geom.o_ones, 1.0f, (int64_t)p->geom.B * p->geom.T * 4, s) != 0)
        return fail(TA3N_ERR_HIP, "fill launch failed");
    HIP_TRY(hipMemcpyAsync(ws + p->geom.o_tuple_first, p->tuple_first.data(), p->tuple_first.size() * sizeof(int32_t),
                           hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));   // tuple_first is pageable host memory owned by the plan
    return TA3N_OK;
}

int ta3n_set_hyper(ta3n_plan *p, float *ws, const ta3n_hyper *h, void *stream) {
    if (!p || !ws || !h) return fail(TA3N_ERR_INVALID, "null argument");
    int rc = ensure_uploaded(p);
    if (rc != TA3N_OK) return rc;
    // by kernel argument: the values are captured when the launch is enqueued, so the host may run any number of steps
    // ahead of the stream (a pinned staging ring would need an event per slot before reuse)
    Hyper hv;
    std::memcpy(&hv, h, sizeof(hv));
    if (launch_set_hyper(ws + p->geom.o_hyper, hv, static_cast<hipStream_t>(stream)) != 0)
        return fail(TA3N_ERR_HIP, std::string("set_hyper launch failed: ") + hipGetErrorString(hipGetLastError()));
    return TA3N_OK;
}

int ta3n_forward(ta3n_plan *p, const float *x, const float *params, float *ws, void *stream) {
    if (!p || !x || !params || !ws) return fail(TA3N_ERR_INVALID, "null argument");
    if (!aligned16(x) || !aligned16(params) || !aligned16(ws)) return fail(TA3N_ERR_INVALID, "buffers must be 16-byte aligned");
    int rc = ensure_uploaded(p);
    if (rc != TA3N_OK) return rc;
    Ptrs ptrs = make_ptrs(p, x, params, nullptr, ws);
    return run_group(p, 0, ptrs, nullptr, nullptr, static_cast<hipStream_t>(stream));
}

int ta3n_loss(ta3n_plan *p, float *ws, void *stream) {
    if (!p || !ws) return fail(TA3N_ERR_INVALID, "null argument");
    int rc = ensure_uploaded(p);
    if (rc != TA3N_OK) return rc;
    Ptrs ptrs = make_ptrs(p, nullptr, nullptr, nullptr, ws);
    return run_group(p, 1, ptrs, nullptr, nullptr, static_cast<hipStream_t>(stream));
}

// ---- ens_DA MCD: the loss assembly around the second classifier and the second, reversed pass (include/ta3n_hip.h) ----
int ta3n_mcd_source_loss(ta3n_plan *p, float *ws, float *scratch, float *out, void *stream) {
    if (!p || !ws || !scratch || !out) return fail(TA3N_ERR_INVALID, "null argument");
    if (!(p->cfg.flags & TA3N_FLAG_MCD) || p->geom.o_Y2 <= 0) return fail(TA3N_ERR_INVALID, "the plan was not created with TA3N_FLAG_MCD");
    if (launch_mcd_source_loss(p->geom, ws, scratch, out, static_cast<hipStream_t>(stream)) != 0) return fail(TA3N_ERR_HIP, "MCD source-loss launch failed");
    return TA3N_OK;
}

int ta3n_mcd_second_loss(ta3n_plan *p, float *ws, float *ws2, int global_target, float *scratch, float *out, void *stream) {
    if (!p || !ws || !ws2 || !scratch || !out || ws == ws2) return fail(TA3N_ERR_INVALID, "null argument (or one workspace for both passes)");
    if (!(p->cfg.flags & TA3N_FLAG_MCD) || p->geom.o_Y2 <= 0) return fail(TA3N_ERR_INVALID, "the plan was not created with TA3N_FLAG_MCD");
    hipStream_t s = static_cast<hipStream_t>(stream);
    // the second pass has no ta3n_loss of its own: every gradient entry its backward reads starts from zero
    for (const char *name : {"gY", "gY2", "gPr", "gPv", "gPf", "g_attn", "gV_ext"})
        for (const auto &r : p->regions)
            if (r.name == name && r.size > 0 && hipMemsetAsync(ws2 + r.off, 0, (size_t)r.size * sizeof(float), s) != hipSuccess)
                return fail(TA3N_ERR_HIP, "memset failed");
    const float inv_count = 1.f / ((float)(global_target > 1 ? global_target : 1) * (float)p->geom.C);      // loss.py:30 torch.mean over (global target videos x classes)
    if (launch_mcd_second_loss(p->geom, ws, ws2, inv_count, scratch, out, s) != 0) return fail(TA3N_ERR_HIP, "MCD second-loss launch failed");
    return TA3N_OK;
}

int ta3n_backward(ta3n_plan *p, const float *x, const float *params, float *grads, float *ws, void *stream) {
    if (!p || !x || !params || !grads || !ws) return fail(TA3N_ERR_INVALID, "null argument");
    if (!aligned16(x) || !aligned16(params) || !aligned16(grads) || !aligned16(ws))
        return fail(TA3N_ERR_INVALID, "buffers must be 16-byte aligned");
    int rc = ensure_uploaded(p);
    if (rc != TA3N_OK) return rc;
    Ptrs ptrs = make_ptrs(p, x, params, grads, ws);
    return run_group(p, 2, ptrs, nullptr, nullptr, static_cast<hipStream_t>(stream));
}

// Per-launch durations with HIP events recorded on `stream` (the stream the kernels
// run on).  Each GEMM / pool / loss phase is launched `reps` times back to back
// between two events (the phases are idempotent); optimiser phases once.
int ta3n_time_phases(ta3n_plan *p, const float *x, float *params, float *grads, float *momentum, float *ws,
                     void *stream, int reps, float *ms_out, int32_t *kind_out, int32_t *group_out, int cap) {
    if (!p || !x || !params || !grads || !momentum || !ws || !ms_out) return fail(TA3N_ERR_INVALID, "null argument");
    int rc = ensure_uploaded(p);
    if (rc != TA3N_OK) return rc;
    if (reps < 1) reps = 1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int n = (int)p->phases.size();
    if (cap < n) return fail(TA3N_ERR_INVALID, "output too small");
    std::vector<hipEvent_t> ev(2 * n);
    for (auto &e : ev) HIP_TRY(hipEventCreate(&e));
    Ptrs ptrs = make_ptrs(p, x, params, grads, ws);
    for (int i = 0; i < n; ++i) {
        const Phase &ph = p->phases[i];
        const int r = (ph.kind == PH_SGD || ph.kind == PH_GRAD_NORM) ? 1 : reps;
        HIP_TRY(hipEventRecord(ev[2 * i], s));
        for (int k = 0; k < r; ++k) {
            int lrc = 0;
            switch (ph.kind) {
                case PH_GEMM: lrc = launch_gemm(ph, static_cast<const Task *>(p->d_tasks), static_cast<const Seg *>(p->d_segs), ptrs, p->geom.o_hyper, p->geom.o_zeros, p->geom.o_ws16, s, nullptr, static_cast<const Wait *>(p->d_waits), p->geom.pair_delta, p->phase_kinds.empty() ? 0 : p->phase_kinds[i]); break;
                case PH_POOL_FWD: lrc = launch_pool_fwd(p->geom, ptrs, s); break;
                case PH_LOSS: lrc = launch_loss(p->geom, ptrs, s); break;
                case PH_POOL_BWD: lrc = launch_pool_bwd(p->geom, ptrs, s); break;
                case PH_HEADS: lrc = launch_heads(p->geom, ptrs, s); break;
                case PH_POOL_CLS: lrc = launch_pool_cls(p->geom, ptrs, s); break;
                case PH_POOL_AVG_FWD: lrc = launch_pool_avg_fwd(p->geom, ptrs, s); break;
                case PH_POOL_AVG_BWD: lrc = launch_pool_avg_bwd(p->geom, ptrs, s); break;
                case PH_BN_FWD: lrc = launch_bn_shared_fwd(p->geom, ptrs, s); break;
                case PH_BN_BWD: lrc = launch_bn_shared_bwd(p->geom, ptrs, s); break;
                case PH_GRAD_NORM: lrc = launch_grad_norm(p->geom, grads, ws, s); break;
                case PH_SGD: lrc = launch_sgd(p->geom, params, grads, momentum, ws, s); break;
                default: lrc = -1;
            }
            if (lrc != 0) {
                for (auto &e : ev) (void)hipEventDestroy(e);
                if (lrc == -5) return TA3N_ERR_INVALID;      // (launch_gemm has set the message: an experiments-only launch list on the default library)
                return fail(TA3N_ERR_HIP, "launch failed while timing");
            }
        }
        HIP_TRY(hipEventRecord(ev[2 * i + 1], s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    for (int i = 0; i < n; ++i) {
        const Phase &ph = p->phases[i];
        const int r = (ph.kind == PH_SGD || ph.kind == PH_GRAD_NORM) ? 1 : reps;
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
        ms_out[i] = ms / (float)r;
        if (kind_out) kind_out[i] = ph.kind;
        if (group_out) group_out[i] = ph.group;
    }
    for (auto &e : ev) (void)hipEventDestroy(e);
    return n;
}

// The two launches that open a pipelined step (ta3n_train_step_after_update): the optimiser update of the shared frame FC,
// and the first GEMM launch carrying the rest of the update as side workgroups.  Unlike ta3n_time_phases this APPLIES the
// update `reps` times (a measurement aid for the end of a benchmark run).
int ta3n_time_update_launches(ta3n_plan *p, const float *x, float *params, float *grads, float *momentum, float *ws, int fused_norm,
                              float lr, float momentum_coef, float weight_decay, float clip, void *stream, int reps, float *ms_out2) {
    if (!p || !x || !params || !grads || !momentum || !ws || !ms_out2) return fail(TA3N_ERR_INVALID, "null argument");
    if (ta3n_has_pipelined_step(p) != 1) return fail(TA3N_ERR_INVALID, "no pipelined step for this configuration");
    int rc = ensure_uploaded(p);
    if (rc != TA3N_OK) return rc;
    if (reps < 1) reps = 1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Geom &g = p->geom;
    hipEvent_t ev[3];
    for (auto &e : ev) HIP_TRY(hipEventCreate(&e));
    Hyper next;
    HIP_TRY(hipMemcpyAsync(&next, ws + g.o_hyper, sizeof(Hyper), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    SgdSide side{params, momentum, lr, momentum_coef, weight_decay, clip, fused_norm ? g.o_sumsq : g.o_norm_part,
                 fused_norm ? g.n_sumsq : g.n_norm_blocks, g.o_p16};
    Ptrs ptrs = make_ptrs(p, x, params, grads, ws);
    HIP_TRY(hipEventRecord(ev[0], s));
    for (int k = 0; k < reps; ++k)
        if (launch_sgd_range(g, params, grads, momentum, ws, 0, p->first_floats, fused_norm != 0, lr, momentum_coef, weight_decay, clip,
                             &next, s) != 0)
            return fail(TA3N_ERR_HIP, "sgd launch failed while timing");
    HIP_TRY(hipEventRecord(ev[1], s));
    for (int k = 0; k < reps; ++k) {
        rc = run_group(p, 5, ptrs, nullptr, nullptr, s, nullptr, 0, 1 << 30, &side);
        if (rc != TA3N_OK) return rc;
    }
    HIP_TRY(hipEventRecord(ev[2], s));
    HIP_TRY(hipStreamSynchronize(s));
    for (int i = 0; i < 2; ++i) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
        ms_out2[i] = ms / (float)reps;
    }
    for (auto &e : ev) (void)hipEventDestroy(e);
    return TA3N_OK;
}

int ta3n_gather_segments(const float *store, const int64_t *first_row, const int32_t *num_frames, const int32_t *labels,
                         const int32_t *video_ids, int n_videos, int num_segments, int feature_dim, float *out,
                         int32_t *labels_out, int32_t *segment_ids_out, void *stream) {
    if (!store || !first_row || !num_frames || !video_ids || !out) return fail(TA3N_ERR_INVALID, "null argument");
    if (labels_out && !labels) return fail(TA3N_ERR_INVALID, "labels_out needs labels");
    if (n_videos < 0 || num_segments <= 0 || feature_dim <= 0) return fail(TA3N_ERR_INVALID, "bad sizes");
    if ((feature_dim & 3) == 0 && (!aligned16(store) || !aligned16(out))) return fail(TA3N_ERR_INVALID, "buffers must be 16-byte aligned");
    if (launch_gather_segments(store, first_row, num_frames, labels, video_ids, n_videos, num_segments, feature_dim, out, labels_out,
                               segment_ids_out, nullptr, static_cast<hipStream_t>(stream)) != 0)
        return fail(TA3N_ERR_HIP, std::string("gather launch failed: ") + hipGetErrorString(hipGetLastError()));
    return TA3N_OK;
}

int ta3n_gather_segments_into(ta3n_plan *p, const float *store, const int64_t *first_row, const int32_t *num_frames,
                              const int32_t *labels, const int32_t *video_ids, int n_videos, int first_video, float *x, float *ws,
                              int32_t *labels_out, void *stream) {
    if (!p || !store || !first_row || !num_frames || !video_ids || !x || !ws) return fail(TA3N_ERR_INVALID, "null argument");
    if (labels_out && !labels) return fail(TA3N_ERR_INVALID, "labels_out needs labels");
    const Geom &g = p->geom;
    if (n_videos < 0 || first_video < 0 || first_video + n_videos > g.B) return fail(TA3N_ERR_INVALID, "videos outside the batch");
    if ((g.D & 3) != 0 || !aligned16(store) || !aligned16(x) || !aligned16(ws)) return fail(TA3N_ERR_INVALID, "feature_dim % 4 and 16-byte alignment required");
    const size_t row0 = (size_t)first_video * g.T;
    float *twin = g.o_x16 >= 0 ? ws + g.o_x16 + row0 * g.D / 2 : nullptr;
    if (launch_gather_segments(store, first_row, num_frames, labels, video_ids, n_videos, g.T, g.D, x + row0 * g.D, labels_out, nullptr,
                               twin, static_cast<hipStream_t>(stream), g.pair_delta) != 0)
        return fail(TA3N_ERR_HIP, std::string("gather launch failed: ") + hipGetErrorString(hipGetLastError()));
    return TA3N_OK;
}

int ta3n_gather_segments_bf16_into(ta3n_plan *p, const void *store16, const int64_t *first_row, const int32_t *num_frames,
                                   const int32_t *labels, const int32_t *video_ids, int n_videos, int first_video, float *x, float *ws,
                                   int32_t *labels_out, void *stream) {
    if (!p || !store16 || !first_row || !num_frames || !video_ids || !ws) return fail(TA3N_ERR_INVALID, "null argument");
    if (labels_out && !labels) return fail(TA3N_ERR_INVALID, "labels_out needs labels");
    const Geom &g = p->geom;
    if (n_videos < 0 || first_video < 0 || first_video + n_videos > g.B) return fail(TA3N_ERR_INVALID, "videos outside the batch");
    if ((g.D & 7) != 0 || !aligned16(store16) || (x && !aligned16(x)) || !aligned16(ws)) return fail(TA3N_ERR_INVALID, "feature_dim % 8 and 16-byte alignment required");
    if (!x && g.o_x16 < 0) return fail(TA3N_ERR_INVALID, "a plan without bf16 twins needs the fp32 input rows (x)");
    const size_t row0 = (size_t)first_video * g.T;
    float *twin = g.o_x16 >= 0 ? ws + g.o_x16 + row0 * g.D / 2 : nullptr;
    if (launch_gather_segments_bf16(store16, first_row, num_frames, labels, video_ids, n_videos, g.T, g.D, x ? x + row0 * g.D : nullptr, labels_out,
                                    twin, static_cast<hipStream_t>(stream), g.pair_delta) != 0)
        return fail(TA3N_ERR_HIP, std::string("gather launch failed: ") + hipGetErrorString(hipGetLastError()));
    return TA3N_OK;
}

int ta3n_eval_metrics(ta3n_plan *p, float *ws, int n_videos, int reset, void *stream) {
    if (!p || !ws) return fail(TA3N_ERR_INVALID, "null argument");
    if (n_videos < 0 || n_videos > p->geom.Bs) return fail(TA3N_ERR_INVALID, "n_videos must be in [0, batch_source]");
    if (p->geom.C > 64) return fail(TA3N_ERR_INVALID, "num_class > 64 not supported");
    if (launch_eval_metrics(p->geom, ws, n_videos, reset, static_cast<hipStream_t>(stream)) != 0)
        return fail(TA3N_ERR_HIP, std::string("metrics launch failed: ") + hipGetErrorString(hipGetLastError()));
    return TA3N_OK;
}

int ta3n_has_fused_step(const ta3n_plan *p) {
    if (!p) return TA3N_ERR_INVALID;
    for (const Phase &ph : p->phases)
        if (ph.group == 4) return 1;
    return 0;
}

int ta3n_train_step(ta3n_plan *p, const float *x, const float *params, float *grads, float *ws, void *stream) {
    return ta3n_train_step_join(p, x, params, grads, ws, stream, nullptr);
}

int ta3n_train_step_range(ta3n_plan *p, const float *x, const float *params, float *grads, float *ws, int first_launch,
                          int n_launches, void *stream) {
    if (!p || !x || !params || !grads || !ws) return fail(TA3N_ERR_INVALID, "null argument");
    if (!aligned16(x) || !aligned16(params) || !aligned16(grads) || !aligned16(ws))
        return fail(TA3N_ERR_INVALID, "buffers must be 16-byte aligned");
    if (ta3n_has_fused_step(p) != 1) return fail(TA3N_ERR_INVALID, "no fused step for this configuration");
    const int total = ta3n_num_phases(p, 4);
    if (first_launch < 0 || n_launches < 0 || first_launch + n_launches > total)
        return fail(TA3N_ERR_INVALID, "launch range outside the fused sequence");
    int rc = ensure_uploaded(p);
    if (rc != TA3N_OK) return rc;
    Ptrs ptrs = make_ptrs(p, x, params, grads, ws);
    return run_group(p, 4, ptrs, nullptr, nullptr, static_cast<hipStream_t>(stream), nullptr, first_launch, n_launches);
}

int ta3n_has_pipelined_step(const ta3n_plan *p) {
    if (!p) return TA3N_ERR_INVALID;
    for (const Phase &ph : p->phases)
        if (ph.group == 5) return 1;
    return 0;
}

int ta3n_train_step_after_update(ta3n_plan *p, const float *x, float *params, float *grads, float *momentum, float *ws,
                                 int fused_norm, float lr, float momentum_coef, float weight_decay, float clip,
                                 const ta3n_hyper *next, void *stream) {
    if (!p || !x || !params || !grads || !momentum || !ws || !next) return fail(TA3N_ERR_INVALID, "null argument");
    if (!aligned16(x) || !aligned16(params) || !aligned16(grads) || !aligned16(momentum) || !aligned16(ws))
        return fail(TA3N_ERR_INVALID, "buffers must be 16-byte aligned");
    if (ta3n_has_pipelined_step(p) != 1) return fail(TA3N_ERR_INVALID, "no pipelined step for this configuration");
    int rc = ensure_uploaded(p);
    if (rc != TA3N_OK) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Geom &g = p->geom;
    if (!fused_norm && launch_grad_norm(g, grads, ws, s) != 0) return fail(TA3N_ERR_HIP, "grad-norm launch failed");
    // (1) the update of the shared frame FC - the only parameters the next launch reads - and the new step's scalars
    if (launch_sgd_range(g, params, grads, momentum, ws, 0, p->first_floats, fused_norm != 0, lr, momentum_coef, weight_decay, clip,
                         reinterpret_cast<const Hyper *>(next), s) != 0)
        return fail(TA3N_ERR_HIP, "sgd launch failed");
    // (2) the new step's first launch, with the rest of the update as side tasks; (3) the other launches of the step
    SgdSide side{params, momentum, lr, momentum_coef, weight_decay, clip, fused_norm ? g.o_sumsq : g.o_norm_part,
                 fused_norm ? g.n_sumsq : g.n_norm_blocks, g.o_p16};
    Ptrs ptrs = make_ptrs(p, x, params, grads, ws);
    rc = run_group(p, 5, ptrs, nullptr, nullptr, s, nullptr, 0, 1 << 30, &side);
    if (rc != TA3N_OK) return rc;
    return run_group(p, 4, ptrs, nullptr, nullptr, s, nullptr, 1, 1 << 30);
}

// One batch assembly of a multi-step call: rows [first_video, first_video + ids_per_step) of the input from a packed store.
static int feed_step(ta3n_plan *p, const ta3n_feed *f, int step, int first_video, int cap, float *x, float *ws, int32_t *labels_out,
                     hipStream_t s) {
    const Geom &g = p->geom;
    if (!f->store || !f->first_row || !f->num_frames || !f->video_ids) return fail(TA3N_ERR_INVALID, "ta3n_feed: null table");
    if (f->ids_per_step < 0 || f->ids_per_step > cap) return fail(TA3N_ERR_INVALID, "ta3n_feed: ids_per_step exceeds the batch half");
    if (labels_out && !f->labels) return fail(TA3N_ERR_INVALID, "ta3n_feed: the source feed needs labels");
    const int32_t *ids = f->video_ids + (size_t)step * f->ids_per_step;
    const size_t row0 = (size_t)first_video * g.T;
    float *twin = g.o_x16 >= 0 ? ws + g.o_x16 + row0 * g.D / 2 : nullptr;
    int rc;
    if (f->bf16) {
        if (!twin && !x) return fail(TA3N_ERR_INVALID, "ta3n_feed: a plan without bf16 twins needs the fp32 input rows");
        rc = launch_gather_segments_bf16(f->store, f->first_row, f->num_frames, f->labels, ids, f->ids_per_step, g.T, g.D,
                                         twin ? nullptr : x + row0 * g.D, labels_out, twin, s, g.pair_delta);
    } else {
        rc = launch_gather_segments(static_cast<const float *>(f->store), f->first_row, f->num_frames, f->labels, ids, f->ids_per_step,
                                    g.T, g.D, x + row0 * g.D, labels_out, nullptr, twin, s, g.pair_delta);
    }
    return rc == 0 ? TA3N_OK : fail(TA3N_ERR_HIP, std::string("gather launch failed: ") + hipGetErrorString(hipGetLastError()));
}

// The same batch half as a job of the launch that opens a pipelined step (launch_sgd_open_feed): same checks, same destinations as feed_step.
static int feed_job(ta3n_plan *p, const ta3n_feed *f, int step, int first_video, int cap, float *x, float *ws, int32_t *labels_out, FeedJob *job) {
    const Geom &g = p->geom;
    if (!f->store || !f->first_row || !f->num_frames || !f->video_ids) return fail(TA3N_ERR_INVALID, "ta3n_feed: null table");
    if (f->ids_per_step < 0 || f->ids_per_step > cap) return fail(TA3N_ERR_INVALID, "ta3n_feed: ids_per_step exceeds the batch half");
    if (labels_out && !f->labels) return fail(TA3N_ERR_INVALID, "ta3n_feed: the source feed needs labels");
    const size_t row0 = (size_t)first_video * g.T;
    float *twin = g.o_x16 >= 0 ? ws + g.o_x16 + row0 * g.D / 2 : nullptr;
    if (f->bf16 && !twin && !x) return fail(TA3N_ERR_INVALID, "ta3n_feed: a plan without bf16 twins needs the fp32 input rows");
    job->store = f->store; job->first_row = f->first_row; job->num_frames = f->num_frames; job->labels = f->labels;
    job->video_ids = f->video_ids + (size_t)step * f->ids_per_step;
    job->n_videos = f->ids_per_step; job->bf16 = f->bf16 ? 1 : 0;
    job->out = (f->bf16 && twin) ? nullptr : x + row0 * g.D;      // (a bf16 store feeding twins writes no fp32 rows: feed_step)
    job->twin = twin; job->labels_out = labels_out;
    return TA3N_OK;
}

// One pipelined step of a multi-step call: optional batch assembly, the update that opens the step (learning rate `lr` of the step
// before, scalars `next` of this one), the step's launches, optional gradient exchange.
static int enqueue_pipelined_step(ta3n_plan *p, const Ptrs &ptrs, float *params, float *grads, float *momentum, float *ws, int fused_norm,
                                  float lr, float momentum_coef, float weight_decay, float clip, const ta3n_hyper *next, int k,
                                  const ta3n_feed *source, const ta3n_feed *target, ta3n_comm *comm, void *scratch_bf16, hipStream_t s) {
    const Geom &g = p->geom;
    int rc;
    // the batch of step k (its input rows are last read by the final launch of step k - 1, already enqueued): assembled by extra workgroups
    // of the launch that opens the step (sgd_open_feed_kernel) - the two gathers as launches of their own cost the step two boundaries
    if (source || target) {
        FeedJob jobs[2];
        std::memset(jobs, 0, sizeof(jobs));
        if (source && (rc = feed_job(p, source, k, 0, g.Bs, const_cast<float *>(ptrs.x), ws, reinterpret_cast<int32_t *>(ws + g.o_labels), &jobs[0])) != TA3N_OK) return rc;
        if (target && (rc = feed_job(p, target, k, g.Bs, g.Bt, const_cast<float *>(ptrs.x), ws, nullptr, &jobs[1])) != TA3N_OK) return rc;
        if (!fused_norm && launch_grad_norm(g, grads, ws, s) != 0) return fail(TA3N_ERR_HIP, "grad-norm launch failed");
        if (launch_sgd_open_feed(g, params, grads, momentum, ws, 0, p->first_floats, fused_norm != 0, lr, momentum_coef, weight_decay, clip,
                                 reinterpret_cast<const Hyper *>(next), jobs, s) != 0)
            return fail(TA3N_ERR_HIP, "sgd + batch-assembly launch failed");
    } else {
        if (!fused_norm && launch_grad_norm(g, grads, ws, s) != 0) return fail(TA3N_ERR_HIP, "grad-norm launch failed");
        if (launch_sgd_range(g, params, grads, momentum, ws, 0, p->first_floats, fused_norm != 0, lr, momentum_coef, weight_decay, clip,
                             reinterpret_cast<const Hyper *>(next), s) != 0)
            return fail(TA3N_ERR_HIP, "sgd launch failed");
    }
    SgdSide side{params, momentum, lr, momentum_coef, weight_decay, clip, fused_norm ? g.o_sumsq : g.o_norm_part,
                 fused_norm ? g.n_sumsq : g.n_norm_blocks, g.o_p16};
    if ((rc = run_group(p, 5, ptrs, nullptr, nullptr, s, nullptr, 0, 1 << 30, &side)) != TA3N_OK) return rc;
    if ((rc = run_group(p, 4, ptrs, nullptr, nullptr, s, nullptr, 1, 1 << 30)) != TA3N_OK) return rc;
    // data parallel: the step's single exchange, on the step's stream, between the last gradient launch and the update
    if (comm && (rc = ta3n_all_reduce_sum(comm, grads, p->live_floats, scratch_bf16, s)) != TA3N_OK) return rc;
    return TA3N_OK;
}

static int check_steps_job(ta3n_plan *p, const float *x, float *params, float *grads, float *momentum, float *ws, const ta3n_hyper *hypers,
                           int fused_norm, const ta3n_feed *source, const ta3n_feed *target, ta3n_comm *comm) {
    if (!p || !x || !params || !grads || !momentum || !ws || !hypers) return fail(TA3N_ERR_INVALID, "null argument");
    if (!aligned16(x) || !aligned16(params) || !aligned16(grads) || !aligned16(momentum) || !aligned16(ws))
        return fail(TA3N_ERR_INVALID, "buffers must be 16-byte aligned");
    if (ta3n_has_pipelined_step(p) != 1) return fail(TA3N_ERR_INVALID, "no pipelined step for this configuration");
    if ((source || target) && (p->geom.D & 7) != 0) return fail(TA3N_ERR_INVALID, "ta3n_feed: feature_dim % 8 required");
    if (comm && fused_norm) return fail(TA3N_ERR_INVALID, "with a communicator the norm is taken from the REDUCED gradients: fused_norm must be 0");
    return ensure_uploaded(p);
}

int ta3n_train_steps(ta3n_plan *p, const float *x, float *params, float *grads, float *momentum, float *ws, int fused_norm,
                     float lr_pending, float momentum_coef, float weight_decay, float clip, const ta3n_hyper *hypers, int n_steps,
                     const ta3n_feed *source, const ta3n_feed *target, ta3n_comm *comm, void *scratch_bf16, void *stream) {
    if (n_steps < 0) return fail(TA3N_ERR_INVALID, "n_steps must be >= 0");
    int rc = check_steps_job(p, x, params, grads, momentum, ws, hypers, fused_norm, source, target, comm);
    if (rc != TA3N_OK) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    Ptrs ptrs = make_ptrs(p, x, params, grads, ws);
    float lr = lr_pending;
    for (int k = 0; k < n_steps; ++k) {
        if ((rc = enqueue_pipelined_step(p, ptrs, params, grads, momentum, ws, fused_norm, lr, momentum_coef, weight_decay, clip, &hypers[k], k,
                                         source, target, comm, scratch_bf16, s)) != TA3N_OK) return rc;
        lr = hypers[k].lr;
    }
    return TA3N_OK;
}

int ta3n_train_steps_multi(const ta3n_steps_job *jobs, int n_jobs, int n_steps) {
    if (!jobs || n_jobs < 1) return fail(TA3N_ERR_INVALID, "null argument");
    if (n_steps < 0) return fail(TA3N_ERR_INVALID, "n_steps must be >= 0");
    std::vector<Ptrs> ptrs;
    for (int j = 0; j < n_jobs; ++j) {
        const ta3n_steps_job &b = jobs[j];
        for (int i = 0; i < j; ++i)
            if (jobs[i].ws == b.ws || jobs[i].grads == b.grads || jobs[i].params == b.params)
                return fail(TA3N_ERR_INVALID, "ta3n_train_steps_multi: jobs must not share buffers");
        int rc = check_steps_job(b.plan, b.x, b.params, b.grads, b.momentum, b.ws, b.hypers, b.fused_norm, b.source, b.target, b.comm);
        if (rc != TA3N_OK) return rc;
        ptrs.push_back(make_ptrs(b.plan, b.x, b.params, b.grads, b.ws));
    }
    // step k of every job before step k + 1 of any: the jobs' queues fill at the same rate, so their launches interleave on the GPU
    for (int k = 0; k < n_steps; ++k)
        for (int j = 0; j < n_jobs; ++j) {
            const ta3n_steps_job &b = jobs[j];
            const int rc = enqueue_pipelined_step(b.plan, ptrs[j], b.params, b.grads, b.momentum, b.ws, b.fused_norm,
                                                  k == 0 ? b.lr_pending : b.hypers[k - 1].lr, b.momentum_coef, b.weight_decay, b.clip,
                                                  &b.hypers[k], k, b.source, b.target, b.comm, b.scratch_bf16, static_cast<hipStream_t>(b.stream));
            if (rc != TA3N_OK) return rc;
        }
    return TA3N_OK;
}

// ---- sharded optimiser step of a data-parallel rank (reduce-scatter -> per-shard norm -> clip + SGD on the own shard -> all-gather) ----
namespace {
// Two regions of the flat prefix, each dealt to the ranks in equal 4-float-aligned chunks: A = the parameters the step's FIRST launch
// reads (shared frame FC; its gradient is the LAST launch's output), B = everything else.  A ends at a multiple of 4 * world at or
// behind first_floats, B may run past live_floats into parameters that never receive a gradient (zeros are reduced, identical values
// are gathered) but never past the buffer.
struct Shards { int64_t a_chunk = 0, a_end = 0, b_chunk = 0, b_end = 0; };
int shard_layout(const ta3n_plan *p, int world, Shards &s) {
    if (!p || world < 1) return fail(TA3N_ERR_INVALID, "bad shard arguments");
    if (p->first_floats <= 0 || p->first_floats >= p->live_floats) return fail(TA3N_ERR_INVALID, "no pipelined step for this configuration");
    const int64_t q = 4 * (int64_t)world;
    s.a_end = (p->first_floats + q - 1) / q * q;
    s.a_chunk = s.a_end / world;
    const int64_t rest = p->live_floats > s.a_end ? p->live_floats - s.a_end : 0;
    s.b_chunk = (rest + q - 1) / q * 4;
    s.b_end = s.a_end + s.b_chunk * world;
    if (s.b_end > p->param_floats) return fail(TA3N_ERR_INVALID, "the flat buffers are too short to shard over this many ranks");
    return TA3N_OK;
}
void own_ranges(const ta3n_plan *p, const Shards &s, int rank, int64_t *r4) {
    r4[0] = std::min<int64_t>(s.a_chunk * rank, p->live_floats);
    r4[1] = std::min<int64_t>(s.a_chunk * (rank + 1), p->live_floats);
    r4[2] = std::min<int64_t>(s.a_end + s.b_chunk * rank, p->live_floats);
    r4[3] = std::min<int64_t>(s.a_end + s.b_chunk * (rank + 1), p->live_floats);
}
int refresh_param_twins(ta3n_plan *p, const float *params, float *ws, int64_t lo, int64_t hi, hipStream_t s) {
    const Geom &g = p->geom;
    if (g.o_p16 < 0 || hi <= lo) return TA3N_OK;
    hi = std::min<int64_t>(hi, p->param_floats);
    int rc;
    if (g.pair_delta) rc = launch_to_bf16_pair(params + lo, ws + g.o_p16 + lo / 2, ws + g.o_p16 + g.pair_delta + lo / 2, hi - lo, s);
    else rc = launch_to_bf16(params + lo, ws + g.o_p16 + lo / 2, hi - lo, s);
    return rc == 0 ? TA3N_OK : fail(TA3N_ERR_HIP, "bf16 conversion launch failed");
}
}  // namespace

int ta3n_shard_ranges(const ta3n_plan *p, int rank, int world, int64_t *own4, int64_t *layout4) {
    Shards s;
    int rc = shard_layout(p, world, s);
    if (rc != TA3N_OK) return rc;
    if (rank < 0 || rank >= world) return fail(TA3N_ERR_INVALID, "rank outside the group");
    if (own4) own_ranges(p, s, rank, own4);
    if (layout4) { layout4[0] = s.a_chunk; layout4[1] = s.a_end; layout4[2] = s.b_chunk; layout4[3] = s.b_end; }
    return TA3N_OK;
}

int ta3n_shard_sumsq(ta3n_plan *p, const float *grads, float *ws, int rank, int world, void *stream) {
    if (!p || !grads || !ws) return fail(TA3N_ERR_INVALID, "null argument");
    Shards s;
    int rc = shard_layout(p, world, s);
    if (rc != TA3N_OK) return rc;
    if (rank < 0 || rank >= world || world > p->geom.n_norm_blocks) return fail(TA3N_ERR_INVALID, "rank outside the group");
    if ((rc = ensure_uploaded(p)) != TA3N_OK) return rc;
    int64_t r[4];
    own_ranges(p, s, rank, r);
    return launch_shard_sumsq(p->geom, grads, ws, r[0], r[1], r[2], r[3], rank, static_cast<hipStream_t>(stream)) == 0
               ? TA3N_OK : fail(TA3N_ERR_HIP, "shard norm launch failed");
}

int ta3n_sgd_shard(ta3n_plan *p, float *params, float *grads, float *momentum, float *ws, int rank, int world, float lr, float momentum_coef,
                   float weight_decay, float clip, const ta3n_hyper *next, void *stream) {
    if (!p || !params || !grads || !momentum || !ws) return fail(TA3N_ERR_INVALID, "null argument");
    Shards s;
    int rc = shard_layout(p, world, s);
    if (rc != TA3N_OK) return rc;
    if (rank < 0 || rank >= world) return fail(TA3N_ERR_INVALID, "rank outside the group");
    if ((rc = ensure_uploaded(p)) != TA3N_OK) return rc;
    int64_t r[4];
    own_ranges(p, s, rank, r);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const Hyper *nx = reinterpret_cast<const Hyper *>(next);
    // the norm is the sum of the ranks' shard partials in region norm_part (slots [0, world), the rest zero): fused_norm = false
    if (r[1] > r[0]) {
        if (launch_sgd_range(p->geom, params, grads, momentum, ws, r[0], r[1], false, lr, momentum_coef, weight_decay, clip, nx, st, true) != 0)
            return fail(TA3N_ERR_HIP, "sgd launch failed");
    } else if (nx) {      // (a rank without a share of region A still needs the next step's scalars)
        if (launch_set_hyper(ws + p->geom.o_hyper, *nx, st) != 0) return fail(TA3N_ERR_HIP, "set_hyper launch failed");
    }
    if (r[3] > r[2] && launch_sgd_range(p->geom, params, grads, momentum, ws, r[2], r[3], false, lr, momentum_coef, weight_decay, clip, nullptr, st,
                                        /* records the norm when the rank has no share of region A */ r[1] <= r[0]) != 0)
        return fail(TA3N_ERR_HIP, "sgd launch failed");
    return TA3N_OK;
}

int ta3n_shard_reduce_scatter(ta3n_plan *p, ta3n_comm *c, float *grads, void *scratch_bf16, void *stream) {
    if (!p || !c || !grads) return fail(TA3N_ERR_INVALID, "null argument");
    Shards s;
    int rc = shard_layout(p, comm_world_size(c), s);
    if (rc != TA3N_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if ((rc = comm_reduce_scatter_sum(c, grads, s.a_end, s.b_chunk, scratch_bf16, st)) != TA3N_OK) return rc;
    return comm_reduce_scatter_sum(c, grads, 0, s.a_chunk, scratch_bf16, st);
}

// everything between "the own shards hold the summed gradients" and "every rank holds the updated parameters", on ONE stream
int ta3n_sharded_update(ta3n_plan *p, ta3n_comm *c, float *params, float *grads, float *momentum, float *ws, float lr, float momentum_coef,
                        float weight_decay, float clip, const ta3n_hyper *next, void *stream) {
    if (!p || !c || !params || !grads || !momentum || !ws) return fail(TA3N_ERR_INVALID, "null argument");
    const int rank = comm_rank(c), world = comm_world_size(c);
    Shards s;
    int rc = shard_layout(p, world, s);
    if (rc != TA3N_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if ((rc = ta3n_shard_sumsq(p, grads, ws, rank, world, stream)) != TA3N_OK) return rc;
    if ((rc = comm_all_gather(c, ws, p->geom.o_norm_part, 1, st)) != TA3N_OK) return rc;
    if ((rc = ta3n_sgd_shard(p, params, grads, momentum, ws, rank, world, lr, momentum_coef, weight_decay, clip, next, stream)) != TA3N_OK) return rc;
    if ((rc = comm_all_gather(c, params, 0, s.a_chunk, st)) != TA3N_OK) return rc;
    if ((rc = comm_all_gather(c, params, s.a_end, s.b_chunk, st)) != TA3N_OK) return rc;
    return refresh_param_twins(p, params, ws, 0, s.b_end, st);
}

int ta3n_train_steps_sharded(ta3n_plan *p, ta3n_comm *c, const float *x, float *params, float *grads, float *momentum, float *ws,
                             float lr_pending, float momentum_coef, float weight_decay, float clip, const ta3n_hyper *hypers, int n_steps,
                             const ta3n_feed *source, const ta3n_feed *target, void *scratch_bf16, void *stream, void *comm_stream) {
    if (!c) return fail(TA3N_ERR_INVALID, "null communicator");
    if (n_steps < 0) return fail(TA3N_ERR_INVALID, "n_steps must be >= 0");
    int rc = check_steps_job(p, x, params, grads, momentum, ws, hypers, 0, source, target, nullptr);
    if (rc != TA3N_OK) return rc;
    const int rank = comm_rank(c), world = comm_world_size(c);
    Shards sh;
    if ((rc = shard_layout(p, world, sh)) != TA3N_OK) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream), cs = static_cast<hipStream_t>(comm_stream);
    const bool two = cs && cs != s;
    hipEvent_t fork = comm_event(c, 0), join = comm_event(c, 1), fork2 = comm_event(c, 2), join2 = comm_event(c, 3);
    if (two && (!fork || !join || !fork2 || !join2)) return fail(TA3N_ERR_HIP, "hipEventCreate failed");
    const Geom &g = p->geom;
    const int n = ta3n_num_phases(p, 4);
    if (n < 3) return fail(TA3N_ERR_INVALID, "no fused step for this configuration");
    // with wgrads_late the last launch also produces gradients of region B: its reduce-scatter then cannot start before that launch
    const bool early_b = two && p->cfg.wgrads_late == 0;
    Ptrs ptrs = make_ptrs(p, x, params, grads, ws);
    float lr = lr_pending;
    for (int k = 0; k < n_steps; ++k) {
        if (source && (rc = feed_step(p, source, k, 0, g.Bs, const_cast<float *>(x), ws, reinterpret_cast<int32_t *>(ws + g.o_labels), s)) != TA3N_OK) return rc;
        if (target && (rc = feed_step(p, target, k, g.Bs, g.Bt, const_cast<float *>(x), ws, nullptr, s)) != TA3N_OK) return rc;
        // -- update of the step before (its gradients are reduce-scattered): norm of the own shards, one float per rank gathered,
        //    clip + Nesterov SGD on the own shards (1 / world of the optimiser pass), carrying this step's scalars
        if ((rc = ta3n_shard_sumsq(p, grads, ws, rank, world, stream)) != TA3N_OK) return rc;
        if ((rc = comm_all_gather(c, ws, g.o_norm_part, 1, s)) != TA3N_OK) return rc;
        if ((rc = ta3n_sgd_shard(p, params, grads, momentum, ws, rank, world, lr, momentum_coef, weight_decay, clip, &hypers[k], stream)) != TA3N_OK) return rc;
        // -- the parameters the first launch reads come back on the step's stream; everything else on the second stream, under that launch
        if ((rc = comm_all_gather(c, params, 0, sh.a_chunk, s)) != TA3N_OK) return rc;
        if ((rc = refresh_param_twins(p, params, ws, 0, sh.a_end, s)) != TA3N_OK) return rc;
        hipStream_t bs = two ? cs : s;
        if (two && (hipEventRecord(fork2, s) != hipSuccess || hipStreamWaitEvent(cs, fork2, 0) != hipSuccess)) return fail(TA3N_ERR_HIP, "event fork failed");
        if ((rc = comm_all_gather(c, params, sh.a_end, sh.b_chunk, bs)) != TA3N_OK) return rc;
        if ((rc = refresh_param_twins(p, params, ws, sh.a_end, sh.b_end, bs)) != TA3N_OK) return rc;
        if (two && hipEventRecord(join2, cs) != hipSuccess) return fail(TA3N_ERR_HIP, "event record failed");
        // -- step k
        if ((rc = run_group(p, 4, ptrs, nullptr, nullptr, s, nullptr, 0, 1)) != TA3N_OK) return rc;
        if (two && hipStreamWaitEvent(s, join2, 0) != hipSuccess) return fail(TA3N_ERR_HIP, "event join failed");
        if ((rc = run_group(p, 4, ptrs, nullptr, nullptr, s, nullptr, 1, n - 2)) != TA3N_OK) return rc;
        if (early_b) {      // region B's gradients are complete: their reduce-scatter runs beside the last launch
            if (hipEventRecord(fork, s) != hipSuccess || hipStreamWaitEvent(cs, fork, 0) != hipSuccess) return fail(TA3N_ERR_HIP, "event fork failed");
            if ((rc = comm_reduce_scatter_sum(c, grads, sh.a_end, sh.b_chunk, scratch_bf16, cs)) != TA3N_OK) return rc;
            if (hipEventRecord(join, cs) != hipSuccess) return fail(TA3N_ERR_HIP, "event record failed");
        }
        if ((rc = run_group(p, 4, ptrs, nullptr, nullptr, s, nullptr, n - 1, 1)) != TA3N_OK) return rc;
        if (!early_b && (rc = comm_reduce_scatter_sum(c, grads, sh.a_end, sh.b_chunk, scratch_bf16, s)) != TA3N_OK) return rc;
        if ((rc = comm_reduce_scatter_sum(c, grads, 0, sh.a_chunk, scratch_bf16, s)) != TA3N_OK) return rc;
        if (early_b && hipStreamWaitEvent(s, join, 0) != hipSuccess) return fail(TA3N_ERR_HIP, "event join failed");
        lr = hypers[k].lr;
    }
    return TA3N_OK;
}

int ta3n_has_fused_update(const ta3n_plan *p) {
    if (!p) return TA3N_ERR_INVALID;
    // every live parameter's gradient is produced by a tile / column-sum task of the fused step (those carry the update)
    // (pair twins: the fused-update epilogue keeps no lo plane of the new parameters - the separate update does)
    // (experiments build only: measured time-neutral in round 3, profiles/r03_fused_update_ab.txt)
    return TA3N_EXPERIMENTS && ta3n_has_fused_step(p) == 1 && p->cfg.aggregation == TA3N_AGG_TRN_M && p->geom.pair_delta == 0 ? 1 : 0;
}

int ta3n_train_steps_fused_update(ta3n_plan *p, const float *x, float *params, float *params_alt, float *grads, float *momentum, float *ws,
                                  float momentum_coef, float weight_decay, float clip, const ta3n_hyper *hypers, int n_steps,
                                  const ta3n_feed *source, const ta3n_feed *target, void *stream) {
    if (!p || !x || !params || !params_alt || !grads || !momentum || !ws || !hypers) return fail(TA3N_ERR_INVALID, "null argument");
    if (n_steps < 0) return fail(TA3N_ERR_INVALID, "n_steps must be >= 0");
    if (!aligned16(x) || !aligned16(params) || !aligned16(params_alt) || !aligned16(grads) || !aligned16(momentum) || !aligned16(ws))
        return fail(TA3N_ERR_INVALID, "buffers must be 16-byte aligned");
    if (ta3n_has_fused_update(p) != 1) return fail(TA3N_ERR_INVALID, "no fused-update step for this configuration");
    int rc = ensure_uploaded(p);
    if (rc != TA3N_OK) return rc;
    if (n_steps == 0) return TA3N_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Geom &g = p->geom;
    if ((source || target) && (g.D & 7) != 0) return fail(TA3N_ERR_INVALID, "ta3n_feed: feature_dim % 8 required");
    const bool twins = g.o_p16 >= 0;
    float *P[2] = {params, params_alt};
    float *T16[2] = {twins ? ws + g.o_p16 : nullptr, twins ? ws + g.o_p16b : nullptr};
    // whatever no tile rewrites (parameters without a gradient in this configuration) must be the same in both buffers
    HIP_TRY(hipMemcpyAsync(params_alt, params, (size_t)p->param_floats * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (twins) HIP_TRY(hipMemcpyAsync(T16[1], T16[0], (size_t)((p->param_floats + 1) / 2) * sizeof(float), hipMemcpyDeviceToDevice, s));
    // the first step's scalars; every later step receives its own from the launch that closes the step before it
    Hyper h0;
    std::memcpy(&h0, &hypers[0], sizeof(h0));
    if (launch_set_hyper(ws + g.o_hyper, h0, s) != 0) return fail(TA3N_ERR_HIP, "set_hyper launch failed");
    int cur = 0;
    for (int k = 0; k < n_steps; ++k) {
        if (source && (rc = feed_step(p, source, k, 0, g.Bs, const_cast<float *>(x), ws, reinterpret_cast<int32_t *>(ws + g.o_labels), s)) != TA3N_OK) return rc;
        if (target && (rc = feed_step(p, target, k, g.Bs, g.Bt, const_cast<float *>(x), ws, nullptr, s)) != TA3N_OK) return rc;
        // forward, heads, backward on P[cur]; every gradient tile writes its block of P[1 - cur] (and its twins)
        Ptrs ptrs = make_ptrs(p, x, P[cur], grads, ws, cur);
        SgdSide side;
        std::memset(&side, 0, sizeof(side));
        side.momentum = momentum; side.lr = hypers[k].lr; side.mu = momentum_coef; side.wd = weight_decay; side.clip = clip;
        side.p16_off = -1; side.p_new = P[1 - cur]; side.p16_new = T16[1 - cur];
        if ((rc = run_group(p, 4, ptrs, nullptr, nullptr, s, nullptr, 0, 1 << 30, &side)) != TA3N_OK) return rc;
        cur ^= 1;
        // norm / clip check of step k on the buffer it wrote, carrying step k + 1's scalars
        if (launch_sgd_fixup(g, P[cur], grads, momentum, ws, T16[cur], hypers[k].lr, momentum_coef, clip,
                             k + 1 < n_steps ? reinterpret_cast<const Hyper *>(&hypers[k + 1]) : nullptr, s) != 0)
            return fail(TA3N_ERR_HIP, "sgd fixup launch failed");
    }
    if (cur == 1) {      // an odd number of steps: the result goes back to the caller's buffer (and its twin region)
        HIP_TRY(hipMemcpyAsync(params, params_alt, (size_t)p->live_floats * sizeof(float), hipMemcpyDeviceToDevice, s));
        if (twins) HIP_TRY(hipMemcpyAsync(T16[0], T16[1], (size_t)((p->param_floats + 1) / 2) * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    return TA3N_OK;
}

int ta3n_refresh_bf16(ta3n_plan *p, const float *x, const float *params, float *ws, void *stream) {
    if (!p || !ws) return fail(TA3N_ERR_INVALID, "null argument");
    if (p->geom.o_ws16 < 0) return TA3N_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc = 0;
    const int64_t nx = (int64_t)p->geom.B * p->geom.T * p->geom.D, pd = p->geom.pair_delta;
    if (pd) {        // pair twins (TA3N_FLAG_F32_SPLIT): the hi and the lo plane
        if (x) rc |= launch_to_bf16_pair(x, ws + p->geom.o_x16, ws + p->geom.o_x16 + pd, nx, s);
        if (params) rc |= launch_to_bf16_pair(params, ws + p->geom.o_p16, ws + p->geom.o_p16 + pd, p->param_floats, s);
    } else {
        if (x) rc |= launch_to_bf16(x, ws + p->geom.o_x16, nx, s);
        if (params) rc |= launch_to_bf16(params, ws + p->geom.o_p16, p->param_floats, s);
    }
    return rc == 0 ? TA3N_OK : fail(TA3N_ERR_HIP, "bf16 conversion launch failed");
}

int ta3n_sgd_range(ta3n_plan *p, float *params, float *grads, float *momentum, float *ws, int64_t begin, int64_t end, int fused_norm,
                   float lr, float momentum_coef, float weight_decay, float clip, void *stream) {
    if (!p || !params || !grads || !momentum || !ws) return fail(TA3N_ERR_INVALID, "null argument");
    if (!aligned16(params) || !aligned16(grads) || !aligned16(momentum)) return fail(TA3N_ERR_INVALID, "buffers must be 16-byte aligned");
    if (begin < 0 || end > p->live_floats || begin > end || (begin & 3) || (end & 3))
        return fail(TA3N_ERR_INVALID, "range must be 4-float aligned and inside the live parameter prefix");
    if (fused_norm && ta3n_has_fused_step(p) != 1) return fail(TA3N_ERR_INVALID, "no fused step for this configuration");
    int rc = ensure_uploaded(p);
    if (rc != TA3N_OK) return rc;
    if (!fused_norm && begin == 0) {
        if (launch_grad_norm(p->geom, grads, ws, static_cast<hipStream_t>(stream)) != 0) return fail(TA3N_ERR_HIP, "grad-norm launch failed");
    }
    if (launch_sgd_range(p->geom, params, grads, momentum, ws, begin, end, fused_norm != 0, lr, momentum_coef, weight_decay, clip,
                         nullptr, static_cast<hipStream_t>(stream)) != 0)
        return fail(TA3N_ERR_HIP, std::string("sgd launch failed: ") + hipGetErrorString(hipGetLastError()));
    return TA3N_OK;
}

int ta3n_sgd_step_next(ta3n_plan *p, float *params, float *grads, float *momentum, float *ws, int fused_norm, float lr,
                       float momentum_coef, float weight_decay, float clip, const ta3n_hyper *next, void *stream) {
    if (!p || !params || !grads || !momentum || !ws || !next) return fail(TA3N_ERR_INVALID, "null argument");
    if (!aligned16(params) || !aligned16(grads) || !aligned16(momentum)) return fail(TA3N_ERR_INVALID, "buffers must be 16-byte aligned");
    if (fused_norm && ta3n_has_fused_step(p) != 1) return fail(TA3N_ERR_INVALID, "no fused step for this configuration");
    static_assert(sizeof(ta3n_hyper) == sizeof(Hyper), "ta3n_hyper and its device mirror differ");
    int rc = ensure_uploaded(p);
    if (rc != TA3N_OK) return rc;
    if (!fused_norm) {
        if (launch_grad_norm(p->geom, grads, ws, static_cast<hipStream_t>(stream)) != 0) return fail(TA3N_ERR_HIP, "grad-norm launch failed");
    }
    if (launch_sgd_range(p->geom, params, grads, momentum, ws, 0, p->live_floats, fused_norm != 0, lr, momentum_coef, weight_decay, clip,
                         reinterpret_cast<const Hyper *>(next), static_cast<hipStream_t>(stream)) != 0)
        return fail(TA3N_ERR_HIP, std::string("sgd launch failed: ") + hipGetErrorString(hipGetLastError()));
    return TA3N_OK;
}

int ta3n_train_step_join(ta3n_plan *p, const float *x, const float *params, float *grads, float *ws, void *stream, void *join_event) {
    if (!p || !x || !params || !grads || !ws) return fail(TA3N_ERR_INVALID, "null argument");
    if (!aligned16(x) || !aligned16(params) || !aligned16(grads) || !aligned16(ws))
        return fail(TA3N_ERR_INVALID, "buffers must be 16-byte aligned");
    if (ta3n_has_fused_step(p) != 1)
        return fail(TA3N_ERR_INVALID, "no fused step for this configuration (needs num_bottleneck == 256, num_class <= 64, fc_dim <= 2048): "
                                      "use ta3n_forward + ta3n_loss + ta3n_backward");
    int rc = ensure_uploaded(p);
    if (rc != TA3N_OK) return rc;
    Ptrs ptrs = make_ptrs(p, x, params, grads, ws);
    return run_group(p, 4, ptrs, nullptr, nullptr, static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(join_event));
}

int ta3n_sgd_step_fused(ta3n_plan *p, float *params, float *grads, float *momentum, float *ws, void *stream) {
    if (!p || !params || !grads || !momentum || !ws) return fail(TA3N_ERR_INVALID, "null argument");
    if (!aligned16(params) || !aligned16(grads) || !aligned16(momentum)) return fail(TA3N_ERR_INVALID, "buffers must be 16-byte aligned");
    if (ta3n_has_fused_step(p) != 1) return fail(TA3N_ERR_INVALID, "no fused step for this configuration: use ta3n_sgd_step");
    int rc = ensure_uploaded(p);
    if (rc != TA3N_OK) return rc;
    if (launch_sgd(p->geom, params, grads, momentum, ws, static_cast<hipStream_t>(stream), true) != 0)
        return fail(TA3N_ERR_HIP, std::string("sgd launch failed: ") + hipGetErrorString(hipGetLastError()));
    return TA3N_OK;
}

int ta3n_sgd_step(ta3n_plan *p, float *params, float *grads, float *momentum, float *ws, void *stream) {
    if (!p || !params || !grads || !momentum || !ws) return fail(TA3N_ERR_INVALID, "null argument");
    if (!aligned16(params) || !aligned16(grads) || !aligned16(momentum)) return fail(TA3N_ERR_INVALID, "buffers must be 16-byte aligned");
    int rc = ensure_uploaded(p);
    if (rc != TA3N_OK) return rc;
    Ptrs ptrs = make_ptrs(p, nullptr, params, grads, ws);
    return run_group(p, 3, ptrs, params, momentum, static_cast<hipStream_t>(stream));
}

}  // extern "C"
