// Host-side launch plan: parameter layout, workspace regions, phases and GEMM
// tile tasks for one (Bs, Bt, T, D, F, C) configuration.
#pragma once
#include <string>
#include <vector>

#include "../../include/ta3n_hip.h"
#include "ta3n_types.h"

struct ParamInfo {
    std::string name;
    int64_t off;
    int32_t rows, cols;   // cols == 0 for 1-D parameters
    bool live;
};

struct Region {
    std::string name;
    int64_t off, size;
};

struct ta3n_plan {
    ta3n_config cfg;
    ta3n::Geom geom;
    std::vector<ParamInfo> params;
    int64_t param_floats = 0, live_floats = 0;
    int64_t first_floats = 0;   // size of the leading parameter block the step's first launch reads (shared frame FC weight + bias)
    std::vector<Region> regions;
    int64_t ws_floats = 0;
    int64_t ws_floats_before_twins = 0;   // (size of the region the ws twins mirror)
    std::vector<ta3n::Seg> segs;
    std::vector<ta3n::Task> tasks;
    std::vector<ta3n::Phase> phases;
    std::vector<ta3n::Wait> waits;      // wait lists of the chained launches (Task.wait_begin / wait_count)
    std::vector<int32_t> tuples, scale_len, scale_id, tuple_first;  // tuple_first[j..j+1) = tuples of scale j
    int n_tuples = 0;
    std::vector<int32_t> phase_kinds;   // per phase: which operand-kind combinations its GEMM tasks use (gemm_tiles' KV bits; filled at upload)
    uint64_t deny_blocking = 0;   // bit i: phase i must not use register-blocked tiles (it does not read bf16 twins; build_plan's retry)
    // device copies (created lazily by the launcher)
    void *d_segs = nullptr;
    void *d_tasks = nullptr;
    void *d_waits = nullptr;
    bool uploaded = false;
    int device = -1;

    int64_t poff(const std::string &name) const;
    int64_t woff(const std::string &name) const;
};

namespace ta3n {
int build_plan(ta3n_plan &p, std::string &err);
void set_error(const std::string &msg);
}
