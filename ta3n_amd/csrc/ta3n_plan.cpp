// Plan builder: turns a ta3n_config into parameter/workspace layouts and the
// per-launch GEMM tile lists.  Pure host code (no HIP calls) so that the CPU
// test-suite can validate the whole wiring against the oracle.
//
// Math being wired (reference file:line in SURVEY.md Appendix A):
//   F1 = drop_i(relu(X Wsh^T + bsh))                           models.py:565-575
//   Pf = Wcd relu(Wfd GRL(F1) + bfd) + bcd                     models.py:456-462
//   Z_t = relu(W_j concat_{f in tau_t} F1[:,f] + b_j)          TRNmodule.py:58-82
//   R_j = sum_{t in scale j} Z_t
//   Pr_j = W2_j relu(W1_j GRL(R_j) + b1_j) + b2_j              models.py:472-488
//   w = 1 - H(softmax Pr);  V = sum_j (1+w_j) R_j;  Vd = drop_v(V)   models.py:351-357, 379-388, 651, 679
//   Y = Wcv Vd + bcv;  Pv = Wcdv relu(Wdv GRL(Vd) + bdv) + bcdv      models.py:686, 464-470
#include "ta3n_plan.h"
#include "ta3n_kernels.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <sstream>

using namespace ta3n;

int64_t ta3n_plan::poff(const std::string &name) const {
    for (const auto &p : params)
        if (p.name == name) return p.off;
    return -1;
}
int64_t ta3n_plan::woff(const std::string &name) const {
    for (const auto &r : regions)
        if (r.name == name) return r.off;
    return -1;
}

namespace {

int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

struct Ref {  // operand reference
    int32_t base, off, ld, kmajor;
};
Ref KC(int32_t base, int64_t off, int32_t ld) { return Ref{base, (int32_t)off, ld, 0}; }  // element (r,k) at off + r*ld + k
Ref KM(int32_t base, int64_t off, int32_t ld) { return Ref{base, (int32_t)off, ld, 1}; }  // element (r,k) at off + k*ld + r

// a_rows > 0: number of readable rows of the A operand when it exceeds the task's output rows (the [K][4] block of
// ones behind the bias-gradient column sums: four identical rows are multiplied, one is stored)
Seg mkseg(Ref a, Ref b, int klen, int scale_kind = SK_ONE, int a_rows = 0) {
    Seg s;
    std::memset(&s, 0, sizeof(s));
    s.pad[0] = a_rows;
    s.a_base = a.base; s.a_off = a.off; s.a_ld = a.ld; s.a_kmajor = a.kmajor;
    s.b_base = b.base; s.b_off = b.off; s.b_ld = b.ld; s.b_kmajor = b.kmajor;
    s.klen = klen;
    s.scale_kind = scale_kind;
    return s;
}

struct GemmSpec {
    int M, N;
    std::vector<Seg> segs;
    Task proto;   // epilogue fields; m0/n0/seg range filled on expansion
    int split = 0;   // 2: two tasks per tile, each over part of the Segs (EPI_SPLITK; the Segs' order is kept: a Seg with a scale stays first)
    int affinity = -1;   // >= 0: specs with the same value read (mostly) the same operand slabs - xcd_aware 3 keeps their tiles on one XCD, back to back
};

Task proto(int32_t c_base, int64_t c_off, int32_t c_ld) {
    Task t;
    std::memset(&t, 0, sizeof(t));
    t.c_base = c_base; t.c_off = (int32_t)c_off; t.c_ld = c_ld;
    t.bias_base = BASE_NONE; t.aux_base = BASE_NONE; t.add_base = BASE_NONE;
    t.alpha_kind = SK_ONE; t.gamma_kind = SK_ONE;
    return t;
}
void with_bias(Task &t, int64_t off) { t.epi |= EPI_BIAS; t.bias_base = BASE_P; t.bias_off = (int32_t)off; }
void with_mask(Task &t, int64_t off, int32_t ld) { t.epi |= EPI_MASK; t.aux_base = BASE_WS; t.aux_off = (int32_t)off; t.aux_ld = ld; }
void with_add(Task &t, int64_t off, int32_t ld) { t.epi |= EPI_ADD; t.add_base = BASE_WS; t.add_off = (int32_t)off; t.add_ld = ld; }

struct Builder {
    ta3n_plan &p;
    explicit Builder(ta3n_plan &pl) : p(pl) {}

    void add_param(const std::string &name, int rows, int cols, bool live) {
        ParamInfo pi;
        pi.name = name; pi.rows = rows; pi.cols = cols; pi.live = live;
        pi.off = p.param_floats;
        p.params.push_back(pi);
        p.param_floats = align_up(p.param_floats + (int64_t)rows * (cols ? cols : 1), 8);   // 8: a bf16 twin row starts 16-byte aligned too
    }
    void add_linear(const std::string &name, int out, int in, bool live) {
        add_param(name + ".weight", out, in, live);
        add_param(name + ".bias", out, 0, live);
    }
    int64_t add_region(const std::string &name, int64_t size) {
        Region r;
        r.name = name; r.off = p.ws_floats; r.size = size;
        p.regions.push_back(r);
        p.ws_floats = align_up(p.ws_floats + size, 64);
        return r.off;
    }

    int n_split_pairs = 0;      // split-K tile pairs handed out so far (EPI_SPLITK)
    int gemm_phase_index = 0;
    int force_next = 0;         // tile code for the next add_gemm_phase only (a launch that mirrors an earlier one)
    // chained launch under construction (begin_chain .. end_chain): the levels' task lists are concatenated into ONE phase whose
    // tile shape is the first level's; chain_level[i] = level of chain_tasks[i]
    bool chaining = false;
    int chain_levels = 0;
    Phase chain_ph;
    std::vector<Task> chain_tasks;
    void begin_chain() { chaining = true; chain_levels = 0; chain_tasks.clear(); }
    void chain_append(std::vector<Task> extra) {
        for (auto &t : extra) { t.sig = -1; t.wait_begin = t.wait_count = 0; chain_tasks.push_back(t); }
    }
    std::string end_chain();    // defined below (derive_chain); returns an error message or ""
    bool mixed_kinds = false;   // a GEMM spec whose Segs differ in operand kinds (not supported by the kernel)
    int sum8[3] = {-1, 0, 0};   // {dst, src, rows}: when dst >= 0 the first workgroup of the next GEMM phase also sums an [rows][8] table
    std::vector<Task> side_tasks;   // non-tile tasks (EPI_COLSUM) appended to the next GEMM phase

    // exact fp32 column sums of a [rows][ld] table of per-workgroup partials -> gradient entries dst[0 .. n): one task per 256 columns
    void colsum_pending(int64_t src, int rows, int ld, int n, int64_t dst) {
        for (int n0 = 0; n0 < n; n0 += 256) {
            Task t;
            std::memset(&t, 0, sizeof(t));
            t.epi = EPI_COLSUM;
            t.c_base = BASE_G; t.c_off = (int32_t)dst; t.c_ld = n;
            t.bias_base = BASE_NONE; t.aux_base = BASE_NONE; t.add_base = BASE_NONE;
            t.m_valid = 1; t.n0 = n0; t.n_valid = std::min(n, n0 + 256);
            t.pad[0] = (int32_t)src; t.pad[1] = rows; t.pad[2] = ld;
            side_tasks.push_back(t);
        }
    }

    // expand GEMM specs into tile tasks of one phase
    void add_gemm_phase(int group, std::vector<GemmSpec> &specs) {
        int wm = 1, wn = 1, wk = 4;
        int forced = p.cfg.tile_config;
        if (chaining && chain_levels > 0) {            // later level of a chained launch: the launch's tile shape is settled
            ++gemm_phase_index;
            forced = chain_ph.wm * 100 + chain_ph.wn * 10 + chain_ph.wk + 1000 * (chain_ph.bf16 & 15) +
                     10000 * ((chain_ph.rm > 1 ? 1 : 0) + (chain_ph.rn > 1 ? 2 : 0));
        } else if (force_next != 0) {
            forced = force_next;
            force_next = 0;
        } else {
            if (gemm_phase_index < 16 && p.cfg.phase_tiles[gemm_phase_index] != 0) forced = p.cfg.phase_tiles[gemm_phase_index];
            ++gemm_phase_index;
        }
        // ten-thousands digit: register blocking (bf16 twins only; dropped again for launches that turn out not to read twins,
        // build_plan's retry loop - deny_blocking)
        int blk = forced / 10000;
        forced %= 10000;
        const bool twins_flags = (p.cfg.flags & TA3N_FLAG_BF16_MFMA) && (p.cfg.flags & TA3N_FLAG_BF16_STORE);
        if (!twins_flags || (p.deny_blocking >> (p.phases.size() & 63)) & 1) blk = 0;
        int forced_stages = forced / 1000;
        forced %= 1000;
        // thousands digit 5 / 6 / 7: HALF stages (64 k per stage: gemm_tiles MODE 5), 2 / 3 / 4 of them - for launches that read plain bf16 twins;
        // dropped again (with the blocking) where a launch turns out not to (build_plan's retry loop)
        bool half_stages = false;
        if (forced_stages >= 5 && forced_stages <= 7) {
            half_stages = twins_flags && !(p.cfg.flags & TA3N_FLAG_F32_SPLIT) && !((p.deny_blocking >> (p.phases.size() & 63)) & 1);
            forced_stages = half_stages ? forced_stages - 3 : 0;
        }
        if (forced != 0) {
            wm = forced / 100; wn = (forced / 10) % 10; wk = forced % 10;
        } else {
            // heuristic from MI355X measurements (tools/proto_gemm.hip, bench.py --autotune):
            // 64x64 tiles with an 8-wave workgroup once they still give every CU a workgroup,
            // otherwise 32x32 tiles, K split 8 ways when there are fewer than two tiles per CU
            auto count = [&](int bm, int bn) {
                int64_t n = 0;
                for (auto &g : specs) n += (int64_t)((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn);
                return n;
            };
            // bf16 twins: 128x128 tiles (2 x 2 blocks per wave) once they still give every CU two workgroups - half the operand
            // bytes per flop through the LDS-DMA (measured at the 512+512-video, 9-segment shape: 153 -> 132 us and 89 -> 73 us
            // for the two long launches; with fewer tiles the 64x64 tile's better balance wins)
            if (twins_flags && !((p.deny_blocking >> (p.phases.size() & 63)) & 1) && count(128, 128) >= 512) { wm = 2; wn = 2; wk = 2; blk = 3; }
            else if (count(64, 64) >= 256) { wm = 2; wn = 2; wk = 2; }
            else if (count(32, 32) >= 512) { wm = 1; wn = 1; wk = 4; }
            else { wm = 1; wn = 1; wk = 8; }
        }
        if (blk >= 4) {
            // 192x128 / 256x128 tiles: half-stage kernels of four waves, and only where every A operand is K-contiguous (forward levels,
            // gradients at activations); anywhere else the launch falls back to the plan's own choice
            bool a_kcontig = true;
            for (auto &g : specs) for (auto &sg : g.segs) a_kcontig = a_kcontig && !sg.a_kmajor;
            if (!(half_stages && forced == 221 && forced_stages == 3 && a_kcontig)) {
                const int64_t n128 = [&] { int64_t n = 0; for (auto &g : specs) n += (int64_t)((g.M + 127) / 128) * ((g.N + 127) / 128); return n; }();
                half_stages = false; forced_stages = 0;
                if (twins_flags && !((p.deny_blocking >> (p.phases.size() & 63)) & 1) && n128 >= 256) { wm = 2; wn = 2; wk = 2; blk = 3; }
                else { wm = 2; wn = 2; wk = 2; blk = 0; }
            }
        }
        const int rm = blk_rm(blk), rn = blk_rn(blk);
        if (forced_stages == 3 && rm * rn == 4 && !half_stages) forced_stages = 2;   // three 64 KiB stages do not fit
        const int BM = 32 * wm * rm, BN = 32 * wn * rn;
        Phase ph;
        std::memset(&ph, 0, sizeof(ph));
        ph.kind = PH_GEMM; ph.group = group; ph.wm = wm; ph.wn = wn; ph.wk = wk; ph.rm = rm; ph.rn = rn;
        if (p.cfg.flags & (TA3N_FLAG_BF16_MFMA | TA3N_FLAG_F32_SPLIT)) {
            // a third stage pays once every tile streams a long K; short-K launches keep the extra resident workgroup
            int min_k = 1 << 30;
            for (auto &g : specs) {
                int k = 0;
                for (auto &sg : g.segs) k += sg.klen;
                min_k = std::min(min_k, k);
            }
            // ... and the twin kernel's third 128-k stage of a 64x64 tile (3 x 32 KB) costs the CU its second resident workgroup.  For the
            // launches of the UNFUSED sequence (ta3n_forward / ta3n_backward: what the module path runs, each launch on its own) two
            // workgroups on two stages each are faster once there is more than one tile per CU (512+512 videos x 9 segments, shared-FC
            // product, 1 152 tiles: 113 -> 86 us; TRN gradient level: 391 -> 260 us; 128+128 x 12, 384 tiles: 29.5 -> 19.7 us), with at
            // most one tile per CU the third stage is (shared-FC weight gradient, 256 tiles: 46.8 vs 53.5 us).  The launches of the
            // pipelined fused step keep the third stage: the same change measured SLOWER there under bench.py's protocol (configs[4]
            // two-stream 0.497 -> 0.504 ms, three alternating processes each: profiles/r04_half_stage_ab.txt) - a launch timed alone
            // and the same launch between its neighbours with the update's side workgroups aboard are different experiments.
            // TA3N_THIRD_STAGE=always / =residency applies one rule to every launch (A/B).
            int64_t n_tiles = 0;
            for (auto &g : specs) n_tiles += (int64_t)((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
            const int64_t twin_stage = (int64_t)(BM + BN) * 256;
            const bool third_costs_a_workgroup = twins_flags && !(p.cfg.flags & TA3N_FLAG_F32_SPLIT) &&
                                                 2 * 3 * twin_stage > 160 * 1024 && 2 * 2 * twin_stage <= 160 * 1024;
            static const int third_rule = [] { const char *e = getenv("TA3N_THIRD_STAGE"); return !e ? 0 : std::strcmp(e, "always") == 0 ? 1 : std::strcmp(e, "residency") == 0 ? 2 : 0; }();
            const bool by_residency = third_rule == 2 || (third_rule == 0 && group < 4);
            const bool third = min_k >= 1024 && rm * rn < 4 && !(by_residency && third_costs_a_workgroup && n_tiles > 256);
            ph.bf16 = forced_stages ? forced_stages : (third ? 3 : 2);
            if (p.cfg.flags & TA3N_FLAG_F32_SPLIT) ph.bf16 |= 32;      // split (hi + lo) operands, three MFMAs per product block
            if (half_stages) ph.bf16 |= 64;
        }
        // (fp32 MFMA kernel: two LDS stages only.  A third stage for the launches with one tile per compute unit - shared-FC product and its
        // weight gradient, 256 tiles at the headline shape - was built and measured in round 5: 257.6 -> 260.3 us per step, 265.7 with
        // three stages everywhere; profiles/r05_persistent_ab.txt.  The thousands digit of a tile code is ignored in the fp32 arithmetic.)
        ph.task_begin = (int32_t)p.tasks.size();
        // A "panel" is the set of tiles of one GEMM that share an operand slab: all
        // column tiles of one row tile when the A side (M*K) is the larger operand,
        // all row tiles of one column tile otherwise.  Panels are dealt to 8 queues
        // (one per XCD: workgroup b runs on XCD b % 8 - a speed assumption only) so
        // the larger operand is partitioned across the private L2s and only the
        // smaller one is replicated.
        struct Panel { std::vector<Task> tiles; int64_t cost; int group; };
        std::vector<Panel> panels;
        int spec_index = -1;
        for (auto &g : specs) {
            ++spec_index;
            const int seg_begin = (int)p.segs.size();
            int cost = 0;
            for (auto &s : g.segs)   // the kernel selects its K loop by the operand kinds of the task's first Seg
                if (s.a_kmajor != g.segs[0].a_kmajor || s.b_kmajor != g.segs[0].b_kmajor ||
                    ((g.proto.epi & EPI_ROWSUM_A) && !(s.a_kmajor && s.b_kmajor)))   // row sums exist in the k-major/k-major loop only
                    mixed_kinds = true;
            for (auto &s : g.segs) {
                p.segs.push_back(s);
                // operands the kernel cannot move 16 bytes at a time take the 4-byte LDS-DMA path: such tiles are
                // several times slower per chunk, so they are weighted up and therefore scheduled first
                const int a_rows = s.pad[0] > 0 ? s.pad[0] : g.M;
                const bool a_vec = s.a_kmajor ? ((s.a_off | s.a_ld | a_rows) & 3) == 0 : ((s.a_off | s.a_ld | s.klen) & 3) == 0;
                const bool b_vec = s.b_kmajor ? ((s.b_off | s.b_ld | g.N) & 3) == 0 : ((s.b_off | s.b_ld | s.klen) & 3) == 0;
                int seg_cost = (s.klen + 63) / 64 * 64 * ((a_vec ? 1 : 3) + (b_vec ? 1 : 3)) / 2;
                // cost_model 1 (measured, profiles/r03_chain_handoff_ab.txt: a weight-gradient tile - both operands k-major - streams
                // a k in ~9 ns, the other kinds in ~5.5 ns; every tile pays ~3 us of descriptor fetch + epilogue whatever its K):
                // the XCD queues are balanced on time, not on K
                if (p.cfg.cost_model == 1 && s.a_kmajor && s.b_kmajor) seg_cost = seg_cost * 17 / 10;
                cost += seg_cost;
            }
            if (p.cfg.cost_model == 1) cost += 576;
            const bool split_m = g.M >= g.N;
            const int outer = split_m ? g.M : g.N, inner = split_m ? g.N : g.M;
            const int bo = split_m ? BM : BN, bi = split_m ? BN : BM;
            for (int o0 = 0; o0 < outer; o0 += bo) {
                Panel pn;
                pn.cost = 0;
                pn.group = g.affinity >= 0 ? g.affinity : -1 - spec_index;      // (no affinity given: the spec's own panels form a group)
                for (int i0 = 0; i0 < inner; i0 += bi) {
                    Task t = g.proto;
                    t.m0 = split_m ? o0 : i0; t.n0 = split_m ? i0 : o0;
                    t.m_valid = g.M; t.n_valid = g.N;
                    t.seg_begin = seg_begin; t.seg_count = (int)g.segs.size();
                    if (t.n0 != 0) t.epi &= ~(uint32_t)EPI_ROWSUM_A;   // the bias gradient is written once per row block
                    t.cost = cost;
                    // split-K: the Segs [0, cut) and [cut, n) go to two tasks; cut = the boundary that balances the two K sums best.
                    // (Not for tiles with a bias gradient or a later Seg that rescales the accumulator: that scale would have to
                    // apply to the other half's contributions too.)
                    bool can_split = g.split == 2 && g.segs.size() >= 2 && !(t.epi & (EPI_ROWSUM_A | EPI_SUMSQ)) && !chaining;
                    for (size_t k = 1; k < g.segs.size(); ++k) can_split = can_split && g.segs[k].scale_kind == SK_ONE;
                    if (can_split) {
                        int total = 0, best_cut = 1, acc_k = 0, best_diff = 1 << 30;
                        for (auto &sg : g.segs) total += sg.klen;
                        for (size_t k = 0; k + 1 < g.segs.size(); ++k) {
                            acc_k += g.segs[k].klen;
                            const int diff = std::abs(2 * acc_k - total);
                            if (diff < best_diff) { best_diff = diff; best_cut = (int)k + 1; }
                        }
                        int k0 = 0;
                        for (int k = 0; k < best_cut; ++k) k0 += g.segs[k].klen;
                        Task h0 = t, h1 = t;
                        h0.epi |= EPI_SPLITK; h1.epi |= EPI_SPLITK;
                        h0.seg_count = best_cut; h1.seg_begin = seg_begin + best_cut; h1.seg_count = (int)g.segs.size() - best_cut;
                        h0.pad[2] = 1; h1.pad[2] = 2;
                        h0.pad[0] = h1.pad[0] = n_split_pairs++;         // (pair id; the ws offsets are assigned once the workspace is laid out)
                        h0.cost = cost * k0 / std::max(total, 1); h1.cost = cost - h0.cost;
                        pn.tiles.push_back(h0); pn.tiles.push_back(h1);
                        pn.cost += cost;
                        continue;
                    }
                    pn.tiles.push_back(t);
                    pn.cost += cost;
                }
                panels.push_back(pn);
            }
        }
        std::vector<Task> local;
        if (p.cfg.xcd_aware == 2) {
            for (auto &pn : panels) for (auto &t : pn.tiles) local.push_back(t);
            std::stable_sort(local.begin(), local.end(), [](const Task &a, const Task &b) { return a.cost > b.cost; });
        } else {
            constexpr int NX = 8;
            std::vector<std::vector<Task>> q(NX);
            std::vector<int64_t> load(NX, 0);
            if (p.cfg.xcd_aware == 3) {
                // Affinity groups (round 6): the panels of specs that share operand slabs - the row panels of one frame's
                // gradient-at-F1 GEMM (same TRN weight slabs), the (scale, position) weight-gradient GEMMs of one scale (same gZ_t), the
                // tuple GEMMs of one scale (same W_j) - stay on ONE XCD and run back to back, so that XCD's L2 fetches the shared slab once
                // for all of them instead of once per panel wherever it landed.  A group heavier than ~0.45 of an XCD's fair share is cut into
                // chunks of consecutive panels; chunks go heaviest first onto the least loaded queue, preferring a queue that already holds
                // a chunk of the same group when that costs less than half a chunk of balance.
                int64_t total = 0;
                for (auto &pn : panels) total += pn.cost;
                const int64_t limit = std::max<int64_t>(1, total * 45 / (100 * NX));
                struct Chunk { std::vector<int> panels; int64_t cost; int group; int64_t tile_cost; };
                std::vector<Chunk> chunks;
                std::vector<int> order;                       // group ids in order of first appearance
                for (auto &pn : panels) if (std::find(order.begin(), order.end(), pn.group) == order.end()) order.push_back(pn.group);
                for (int gid : order) {
                    Chunk ck{{}, 0, gid, 0};
                    for (int i = 0; i < (int)panels.size(); ++i) {
                        if (panels[i].group != gid) continue;
                        if (!ck.panels.empty() && ck.cost + panels[i].cost > limit) { chunks.push_back(ck); ck = Chunk{{}, 0, gid, 0}; }
                        ck.panels.push_back(i);
                        ck.cost += panels[i].cost;
                        for (auto &t : panels[i].tiles) ck.tile_cost = std::max<int64_t>(ck.tile_cost, t.cost);
                    }
                    if (!ck.panels.empty()) chunks.push_back(ck);
                }
                std::stable_sort(chunks.begin(), chunks.end(), [](const Chunk &a, const Chunk &b) { return a.cost > b.cost; });
                std::vector<std::vector<int>> held(NX);       // chunk indices per queue
                for (int ci = 0; ci < (int)chunks.size(); ++ci) {
                    int best = 0;
                    for (int x = 1; x < NX; ++x) if (load[x] < load[best]) best = x;
                    for (int x = 0; x < NX; ++x) {
                        bool same = false;
                        for (int h : held[x]) same = same || chunks[h].group == chunks[ci].group;
                        if (same && load[x] <= load[best] + chunks[ci].cost / 2) { best = x; break; }
                    }
                    held[best].push_back(ci);
                    load[best] += chunks[ci].cost;
                }
                for (int x = 0; x < NX; ++x) {                // per queue: chunks with the longest tiles first, a group's chunks adjacent
                    std::stable_sort(held[x].begin(), held[x].end(), [&](int a, int b) {
                        if (chunks[a].tile_cost != chunks[b].tile_cost) return chunks[a].tile_cost > chunks[b].tile_cost;
                        return chunks[a].group < chunks[b].group;
                    });
                    for (int h : held[x]) for (int i : chunks[h].panels) for (auto &t : panels[i].tiles) q[x].push_back(t);
                }
            } else {
                std::stable_sort(panels.begin(), panels.end(), [](const Panel &a, const Panel &b) { return a.cost > b.cost; });
                for (auto &pn : panels) {   // heaviest panel first onto the least loaded XCD queue
                    int best = 0;
                    for (int x = 1; x < NX; ++x) if (load[x] < load[best]) best = x;
                    for (auto &t : pn.tiles) q[best].push_back(t);
                    load[best] += pn.cost;
                }
                for (auto &v : q) std::stable_sort(v.begin(), v.end(), [](const Task &a, const Task &b) { return a.cost > b.cost; });
            }
            size_t depth = 0;
            for (auto &v : q) depth = std::max(depth, v.size());
            Task nop;
            std::memset(&nop, 0, sizeof(nop));   // seg_count == 0: the workgroup exits immediately
            nop.c_base = BASE_NONE; nop.bias_base = BASE_NONE; nop.aux_base = BASE_NONE; nop.add_base = BASE_NONE;
            // (affinity groups leave queues of very different LENGTH - one of many short tiles beside seven of few long ones.  Padding the
            // short queues with empty workgroups keeps "workgroup b runs on XCD b % 8" exact, but only pays while most queues still have
            // tiles: once fewer than half do, the rest is appended unpadded and spreads over all XCDs - short tiles, balance over locality.)
            size_t padded_depth = depth;
            if (p.cfg.xcd_aware == 3) {
                std::vector<size_t> sizes;
                for (auto &v : q) sizes.push_back(v.size());
                std::sort(sizes.begin(), sizes.end());
                padded_depth = sizes[NX / 2];      // depth at which half of the queues have run out
            }
            for (size_t d = 0; d < padded_depth; ++d)
                for (int x = 0; x < NX; ++x) local.push_back(d < q[x].size() ? q[x][d] : nop);
            while (!local.empty() && local.back().seg_count == 0) local.pop_back();
            for (size_t d = padded_depth; d < depth; ++d)
                for (int x = 0; x < NX; ++x) if (d < q[x].size()) local.push_back(q[x][d]);
        }
        if (sum8[0] >= 0 && !local.empty()) {
            // on the LAST task of the list - a real tile (trailing padding was popped) and the launch's lightest, so the side job's
            // dependent loads sit under the other tiles.  (Until round 5 it rode on local[0], the HEAVIEST tile: ~1.5 us on the critical
            // path of the relation-level backward launch.)
            Task &host = local.back();
            host.epi |= EPI_SUMROWS8;
            host.pad[0] = sum8[0]; host.pad[1] = sum8[1]; host.pad[2] = sum8[2];
            sum8[0] = -1;
        }
        // the short side tasks go right behind the first tile: they finish under the tiles instead of extending the launch's tail
        local.insert(local.begin() + (local.empty() ? 0 : 1), side_tasks.begin(), side_tasks.end());
        side_tasks.clear();
        ph.chain_off = -1; ph.chain_n = 0;
        if (chaining) {
            if (chain_levels == 0) { chain_ph = ph; chain_ph.group = group; }
            ++chain_levels;
            for (auto &t : local) { t.sig = -1; chain_tasks.push_back(t); }
            return;
        }
        for (auto &t : local) { t.sig = -1; p.tasks.push_back(t); }
        ph.task_count = (int32_t)local.size();
        p.phases.push_back(ph);
    }
    void sum8_pending(int dst, int src, int rows) { sum8[0] = dst; sum8[1] = src; sum8[2] = rows; }
    void add_simple_phase(int kind, int group) {
        Phase ph;
        std::memset(&ph, 0, sizeof(ph));
        ph.kind = kind; ph.group = group;
        p.phases.push_back(ph);
    }
};


// ---------------------------------------------------------------------------------------------------------------------
// Chained launches: who waits for whom.  Every task's writes and reads are laid out as element intervals per buffer (the
// twins mirror their originals element for element, so this runs on the original offsets, before add_bf16_twins); a task
// depends on every EARLIER task of the launch whose writes overlap its reads.  Producers with the same set of consumers share
// one counter (target = their number), a consumer's wait list names the counters of its producers.  A later task overwriting
// what an earlier one reads or writes would be a race inside one launch: reported as an error (such levels must stay launches).
struct Ival { int64_t lo, hi; int task; };

static void tile_rows(std::vector<Ival> &out, int64_t off, int64_t ld, int r0, int nr, int c0, int nc, int task) {
    for (int r = r0; r < r0 + nr; ++r) out.push_back(Ival{off + (int64_t)r * ld + c0, off + (int64_t)r * ld + c0 + nc, task});
}

std::string Builder::end_chain() {
    chaining = false;
    Phase ph = chain_ph;
    const int BM = 32 * ph.wm * std::max(ph.rm, 1), BN = 32 * ph.wn * std::max(ph.rn, 1);
    const int n = (int)chain_tasks.size();
    std::vector<Ival> wr[BASE_COUNT], rd[BASE_COUNT];
    auto operand = [&](std::vector<Ival> &out, int off, int ld, int kmajor, int r0, int rows_valid, int tile_rows_n, int klen, int task) {
        const int nr = std::min(tile_rows_n, rows_valid - r0);
        if (nr <= 0) return;
        if (!kmajor) tile_rows(out, off, ld, r0, nr, 0, klen, task);
        else for (int k = 0; k < klen; ++k) out.push_back(Ival{off + (int64_t)k * ld + r0, off + (int64_t)k * ld + r0 + nr, task});
    };
    for (int i = 0; i < n; ++i) {
        const Task &t = chain_tasks[i];
        if (t.epi & EPI_SGD) {            // optimiser side job: writes the parameters [4 pad0, 4 pad1) (reads gradients / norm partials of earlier launches)
            wr[BASE_P].push_back(Ival{4ll * t.pad[0], 4ll * t.pad[1], i});
            continue;
        }
        if (t.epi & EPI_COLSUM) {
            tile_rows(rd[BASE_WS], t.pad[0], t.pad[2], 0, t.pad[1], t.n0, t.n_valid - t.n0, i);
            wr[t.c_base].push_back(Ival{(int64_t)t.c_off + t.n0, (int64_t)t.c_off + t.n_valid, i});
            if (t.epi & EPI_SUMSQ) wr[BASE_WS].push_back(Ival{t.pad[3], t.pad[3] + 1, i});
            continue;
        }
        if (t.seg_count == 0) continue;
        if (t.epi & EPI_SUMROWS8) {
            rd[BASE_WS].push_back(Ival{t.pad[1], t.pad[1] + 8ll * t.pad[2], i});
            wr[BASE_WS].push_back(Ival{t.pad[0], t.pad[0] + 8, i});
        }
        const int nr = std::min(BM, t.m_valid - t.m0), nc = std::min(BN, t.n_valid - t.n0);
        for (int k = t.seg_begin; k < t.seg_begin + t.seg_count; ++k) {
            const Seg &sg = p.segs[k];
            operand(rd[sg.a_base], sg.a_off, sg.a_ld, sg.a_kmajor, t.m0, sg.pad[0] > 0 ? sg.pad[0] : t.m_valid, BM, sg.klen, i);
            operand(rd[sg.b_base], sg.b_off, sg.b_ld, sg.b_kmajor, t.n0, t.n_valid, BN, sg.klen, i);
        }
        if (t.epi & EPI_BIAS) rd[t.bias_base].push_back(Ival{(int64_t)t.bias_off + t.n0, (int64_t)t.bias_off + t.n0 + nc, i});
        if (t.epi & EPI_MASK) tile_rows(rd[t.aux_base], t.aux_off, t.aux_ld, t.m0, nr, t.n0, nc, i);
        if (t.epi & EPI_ADD) tile_rows(rd[t.add_base], t.add_off, t.add_ld, t.m0, nr, t.n0, nc, i);
        for (int f = 0; f < t.fan_count; ++f) {
            tile_rows(rd[BASE_WS], t.fan_mask_off[f], t.fan_ld, t.m0, nr, t.n0, nc, i);
            tile_rows(wr[BASE_WS], t.fan_out_off[f], t.fan_ld, t.m0, nr, t.n0, nc, i);
        }
        tile_rows(wr[t.c_base], t.c_off, t.c_ld, t.m0, nr, t.n0, nc, i);
        if (t.epi & EPI_ROWSUM_A) wr[t.bias_base].push_back(Ival{(int64_t)t.bias_off + t.m0, (int64_t)t.bias_off + t.m0 + nr, i});
        if (t.epi & EPI_SUMSQ) wr[BASE_WS].push_back(Ival{t.pad[3], t.pad[3] + 1, i});
    }
    std::vector<std::vector<int>> consumers(n), producers(n);
    for (int b = 0; b < BASE_COUNT; ++b) {
        auto &w = wr[b];
        std::sort(w.begin(), w.end(), [](const Ival &a, const Ival &c) { return a.lo < c.lo; });
        for (size_t k = 1; k < w.size(); ++k)
            if (w[k].lo < w[k - 1].hi && w[k].task != w[k - 1].task)
                return "chained launch: tasks " + std::to_string(w[k - 1].task) + " and " + std::to_string(w[k].task) + " write the same elements";
        for (const Ival &r : rd[b]) {
            // writers are sorted and disjoint: the first one that can overlap ends after r.lo
            size_t k = std::lower_bound(w.begin(), w.end(), r.lo, [](const Ival &a, int64_t v) { return a.hi <= v; }) - w.begin();
            for (; k < w.size() && w[k].lo < r.hi; ++k) {
                if (w[k].task == r.task) continue;
                if (w[k].task > r.task)
                    return "chained launch: task " + std::to_string(w[k].task) + " overwrites what the earlier task " + std::to_string(r.task) + " reads";
                producers[r.task].push_back(w[k].task);
            }
        }
    }
    for (int i = 0; i < n; ++i) {
        auto &v = producers[i];
        std::sort(v.begin(), v.end());
        v.erase(std::unique(v.begin(), v.end()), v.end());
        for (int j : v) consumers[j].push_back(i);      // (i ascending: the lists come out sorted)
    }
    std::vector<std::pair<std::vector<int>, int>> groups;   // (consumer set, counter)
    std::vector<int> target;
    for (int j = 0; j < n; ++j) {
        if (consumers[j].empty()) continue;
        int c = -1;
        for (auto &gp : groups)
            if (gp.first == consumers[j]) { c = gp.second; break; }
        if (c < 0) { c = (int)target.size(); groups.push_back({consumers[j], c}); target.push_back(0); }
        ++target[c];
        chain_tasks[j].sig = c;
    }
    for (int i = 0; i < n; ++i) {
        std::vector<int> cs;
        for (int j : producers[i]) cs.push_back(chain_tasks[j].sig);
        std::sort(cs.begin(), cs.end());
        cs.erase(std::unique(cs.begin(), cs.end()), cs.end());
        chain_tasks[i].wait_begin = (int32_t)p.waits.size();
        chain_tasks[i].wait_count = (int32_t)cs.size();
        for (int c : cs) p.waits.push_back(Wait{c, target[c]});
    }
    ph.chain_n = (int32_t)target.size();
    ph.chain_off = (int32_t)add_region("chain" + std::to_string(p.phases.size()), 2 + ph.chain_n);
    ph.task_begin = (int32_t)p.tasks.size();
    ph.task_count = n;
    for (auto &t : chain_tasks) p.tasks.push_back(t);
    p.phases.push_back(ph);
    chain_tasks.clear();
    return "";
}

typedef std::pair<int64_t, int64_t> Span;   // [first, last) in ws floats

// bf16 twins (TA3N_FLAG_BF16_STORE).  extra_produced: ws spans whose twin a non-GEMM kernel of the fused step keeps current.
void add_bf16_twins(ta3n_plan &p, Builder &b, Geom &g, int BT, int D, const std::vector<Span> &extra_produced,
                    const std::vector<Span> &gemm_only = {}) {
    const ta3n_config &c = p.cfg;
    // (TA3N_FLAG_F32_SPLIT | _BF16_STORE: "pair twins" - every twin has a hi and a lo plane, x = hi + lo to 16 mantissa bits)
    if (!((c.flags & (TA3N_FLAG_BF16_MFMA | TA3N_FLAG_F32_SPLIT)) && (c.flags & TA3N_FLAG_BF16_STORE))) return;
    // bf16 twins.  A launch of the fused step reads twins when every one of its operands can be moved 16 bytes (8
    // elements) at a time and has a producer that keeps the twin current; its Segs are then re-addressed into the
    // twin regions.  Launches with odd-shaped operands (the small head weight gradients) keep rounding fp32
    // operands in registers.
    g.ws16_span = (int32_t)p.ws_floats;
    p.ws_floats_before_twins = p.ws_floats;
    g.o_ws16 = (int32_t)b.add_region("ws16", (p.ws_floats + 1) / 2);
    g.o_p16 = (int32_t)b.add_region("p16", (p.param_floats + 1) / 2);
    g.o_x16 = (int32_t)b.add_region("x16", ((int64_t)BT * D + 1) / 2);
    g.o_p16b = (int32_t)b.add_region("p16b", (p.param_floats + 1) / 2);      // (fused-update step: twins of the second parameter buffer)
    if (c.flags & TA3N_FLAG_F32_SPLIT) {
        // the lo planes: the same four regions again, so ONE displacement leads from any twin element to its lo half
        const int32_t lo_ws = (int32_t)b.add_region("ws16_lo", (p.ws_floats_before_twins + 1) / 2);
        const int32_t lo_p = (int32_t)b.add_region("p16_lo", (p.param_floats + 1) / 2);
        const int32_t lo_x = (int32_t)b.add_region("x16_lo", ((int64_t)BT * D + 1) / 2);
        const int32_t lo_pb = (int32_t)b.add_region("p16b_lo", (p.param_floats + 1) / 2);
        g.pair_delta = lo_ws - g.o_ws16;
        if (lo_p - g.o_p16 != g.pair_delta || lo_x - g.o_x16 != g.pair_delta || lo_pb - g.o_p16b != g.pair_delta || (g.pair_delta & 3))
            g.pair_delta = -1;      // (cannot happen: equal sizes, 64-float alignment; build_plan reports it)
    }
    auto twin = [&](int32_t &base, int32_t &off) {
        if (base == BASE_P) { base = BASE_P16; off = off / 2; return; }      // relative to the twin region the launch is handed
        const int32_t origin = base == BASE_WS ? g.o_ws16 : g.o_x16;
        off = origin + off / 2;
        base = BASE_WS;
    };
    auto overlaps = [](const Span &a, const Span &b) { return a.first < b.second && b.first < a.second; };
    // Two launch sequences share the workspace and its twin regions: the FUSED step (groups 4 / 5) and - round 6 - the UNFUSED lists
    // (groups 0 / 2: ta3n_forward / ta3n_backward, what the DA options with a loss term between forward and backward run; their GEMM launches
    // were twice as long on fp32 stages rounded in registers as the fused step's on twins).  Each family is analysed on its own: who keeps a
    // twin up to date, which launch may read twins, which producers store them.  (TA3N_FLAG_MCD: the second pass runs on a second workspace -
    // its caller copies the parameter / input twins over from the first before it, TrainEngine.mcd_second_forward.)
    auto fused_family = [](int group) { return group == 4 || group == 5; };
    auto unfused_family = [](int group) { return group == 0 || group == 2; };
    auto analyse = [&](bool (*in_family)(int), const std::vector<Span> &extra) {
    // who keeps a twin up to date: GEMM tiles of the family (their C and their fan-out copies) and - fused step - the heads
    // kernel for gHf.  A launch may read twins only of such data (plus parameters and the input).
    std::vector<Span> produced;   // (Span = [first, last) in ws floats)
    for (auto &sp : extra) produced.push_back(sp);
    for (const Phase &ph : p.phases) {
        if (!in_family(ph.group) || ph.kind != PH_GEMM) continue;
        for (int i = ph.task_begin; i < ph.task_begin + ph.task_count; ++i) {
            const Task &t = p.tasks[i];
            if (t.seg_count == 0) continue;
            if (t.c_base == BASE_WS) produced.push_back({t.c_off, t.c_off + (int64_t)(t.m_valid - 1) * t.c_ld + t.n_valid});
            for (int f = 0; f < t.fan_count; ++f)
                produced.push_back({t.fan_out_off[f], t.fan_out_off[f] + (int64_t)(t.m_valid - 1) * t.fan_ld + t.n_valid});
        }
    }
    std::vector<Span> read16;   // ws spans some twin-reading Seg covers
    for (Phase &ph : p.phases) {
        if (!in_family(ph.group) || ph.kind != PH_GEMM) continue;
        bool ok = true;
        std::vector<Span> reads;
        std::vector<char> seen(p.segs.size(), 0);
        for (int i = ph.task_begin; i < ph.task_begin + ph.task_count && ok; ++i) {
            const Task &t = p.tasks[i];
            for (int k = t.seg_begin; k < t.seg_begin + t.seg_count && ok; ++k) {
                if (seen[k]) continue;
                seen[k] = 1;
                const Seg &sg = p.segs[k];
                const int a_rows = sg.pad[0] > 0 ? sg.pad[0] : t.m_valid, b_rows = t.n_valid;
                auto side_ok = [&](int base, int off, int ld, int kmajor, int rows) {
                    if (base == BASE_G) return false;
                    if (((off | ld) & 7) != 0) return false;
                    if ((kmajor ? rows : sg.klen) & 7) return false;      // the 16-byte pieces run along rows (k-major) or k
                    if (base == BASE_WS) {
                        const Span rd = kmajor ? Span{off, off + (int64_t)(sg.klen - 1) * ld + rows}
                                               : Span{off, off + (int64_t)(rows - 1) * ld + sg.klen};
                        bool covered = false;
                        for (auto &pr : produced) covered = covered || overlaps(rd, pr);
                        if (!covered) return false;
                        reads.push_back(rd);
                    }
                    return true;
                };
                ok = side_ok(sg.a_base, sg.a_off, sg.a_ld, sg.a_kmajor, a_rows) &&
                     side_ok(sg.b_base, sg.b_off, sg.b_ld, sg.b_kmajor, b_rows);
            }
        }
        if (!ok) continue;
        ph.bf16 |= 16;
        read16.insert(read16.end(), reads.begin(), reads.end());
        std::fill(seen.begin(), seen.end(), 0);
        for (int i = ph.task_begin; i < ph.task_begin + ph.task_count; ++i) {
            const Task &t = p.tasks[i];
            for (int k = t.seg_begin; k < t.seg_begin + t.seg_count; ++k) {
                if (seen[k]) continue;
                seen[k] = 1;
                twin(p.segs[k].a_base, p.segs[k].a_off);
                twin(p.segs[k].b_base, p.segs[k].b_off);
            }
        }
    }
    // producers whose output some twin-reading Seg covers store the twin as well
    for (const Phase &ph : p.phases) {
        if (!in_family(ph.group) || ph.kind != PH_GEMM) continue;
        for (int i = ph.task_begin; i < ph.task_begin + ph.task_count; ++i) {
            Task &t = p.tasks[i];
            if (t.seg_count == 0) continue;
            if (t.c_base == BASE_WS) {
                const Span out{t.c_off, t.c_off + (int64_t)(t.m_valid - 1) * t.c_ld + t.n_valid};
                for (auto &rd : read16)
                    if (overlaps(out, rd)) { t.epi |= EPI_TWIN16; break; }
            }
            for (int f = 0; f < t.fan_count; ++f) {
                const Span out{t.fan_out_off[f], t.fan_out_off[f] + (int64_t)(t.m_valid - 1) * t.fan_ld + t.n_valid};
                for (auto &rd : read16)
                    if (overlaps(out, rd)) { t.epi |= EPI_TWIN16_FAN; break; }
            }
        }
    }
    };
    analyse(fused_family, extra_produced);
    const char *ue = std::getenv("TA3N_UNFUSED_TWINS");      // (=0: the unfused lists on fp32 stages rounded in registers, as before round 6 - A/B aid)
    if (!(ue && std::atoi(ue) == 0)) analyse(unfused_family, {});
    // gemm_only: workspace regions that only GEMM launches read (no pointwise kernel, no API output).  If every launch
    // that reads such a region reads its twin, the producers skip the fp32 store (EPI_TWIN_ONLY): the fp32 region then
    // holds nothing meaningful in this configuration.
    for (const Span &reg : gemm_only) {
        bool fp32_reader = false;
        for (const Phase &ph : p.phases) {
            if ((ph.group != 4 && ph.group != 5) || ph.kind != PH_GEMM || (ph.bf16 & 16)) continue;
            for (int i = ph.task_begin; i < ph.task_begin + ph.task_count; ++i) {
                const Task &t = p.tasks[i];
                for (int k = t.seg_begin; k < t.seg_begin + t.seg_count; ++k) {
                    const Seg &sg = p.segs[k];
                    auto hit = [&](int base, int off, int ld, int kmajor, int rows) {
                        if (base != BASE_WS) return false;
                        const Span rd = kmajor ? Span{off, off + (int64_t)(sg.klen - 1) * ld + rows} : Span{off, off + (int64_t)(rows - 1) * ld + sg.klen};
                        return overlaps(rd, reg);
                    };
                    fp32_reader = fp32_reader || hit(sg.a_base, sg.a_off, sg.a_ld, sg.a_kmajor, sg.pad[0] > 0 ? sg.pad[0] : t.m_valid) ||
                                  hit(sg.b_base, sg.b_off, sg.b_ld, sg.b_kmajor, t.n_valid);
                }
                if ((t.epi & (EPI_MASK | EPI_ADD)) && t.seg_count > 0) {     // epilogue operands are read in fp32
                    if (t.aux_base == BASE_WS && (t.epi & EPI_MASK) && overlaps(Span{t.aux_off, t.aux_off + (int64_t)(t.m_valid - 1) * t.aux_ld + t.n_valid}, reg)) fp32_reader = true;
                    if (t.add_base == BASE_WS && (t.epi & EPI_ADD) && overlaps(Span{t.add_off, t.add_off + (int64_t)(t.m_valid - 1) * t.add_ld + t.n_valid}, reg)) fp32_reader = true;
                }
            }
        }
        // (twin launches read masks / residuals in fp32 too)
        for (const Phase &ph : p.phases) {
            if ((ph.group != 4 && ph.group != 5) || ph.kind != PH_GEMM || !(ph.bf16 & 16)) continue;
            for (int i = ph.task_begin; i < ph.task_begin + ph.task_count; ++i) {
                const Task &t = p.tasks[i];
                if (t.seg_count == 0) continue;
                if ((t.epi & EPI_MASK) && t.aux_base == BASE_WS && overlaps(Span{t.aux_off, t.aux_off + (int64_t)(t.m_valid - 1) * t.aux_ld + t.n_valid}, reg)) fp32_reader = true;
                if ((t.epi & EPI_ADD) && t.add_base == BASE_WS && overlaps(Span{t.add_off, t.add_off + (int64_t)(t.m_valid - 1) * t.add_ld + t.n_valid}, reg)) fp32_reader = true;
                for (int f = 0; f < t.fan_count; ++f)
                    if (overlaps(Span{t.fan_mask_off[f], t.fan_mask_off[f] + (int64_t)(t.m_valid - 1) * t.fan_ld + t.n_valid}, reg)) fp32_reader = true;
            }
        }
        if (fp32_reader) continue;
        for (const Phase &ph : p.phases) {
            if ((ph.group != 4 && ph.group != 5) || ph.kind != PH_GEMM) continue;
            for (int i = ph.task_begin; i < ph.task_begin + ph.task_count; ++i) {
                Task &t = p.tasks[i];
                if (t.seg_count == 0) continue;
                if ((t.epi & EPI_TWIN16) && t.c_base == BASE_WS &&
                    overlaps(Span{t.c_off, t.c_off + (int64_t)(t.m_valid - 1) * t.c_ld + t.n_valid}, reg)) t.epi |= EPI_TWIN_ONLY;
                if (t.epi & EPI_TWIN16_FAN) {
                    bool all_in = t.fan_count > 0;
                    for (int f = 0; f < t.fan_count; ++f)
                        all_in = all_in && overlaps(Span{t.fan_out_off[f], t.fan_out_off[f] + (int64_t)(t.m_valid - 1) * t.fan_ld + t.n_valid}, reg);
                    if (all_in) t.epi |= EPI_TWIN_ONLY_FAN;
                }
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// TA3N_AGG_AVGPOOL: BASELINE configs[0] (TemPooling, source-only).  Reference: models.py:557-579 (shared frame FC),
// 421-433 (aggregate_frames, "1. averaging"), 679-687 (dropout_v, classifier), main.py:439-451 (CE on the source rows),
// 574-583 (backward, clip, SGD).  Launches: F1 GEMM, pool_cls kernel, {dWsh, dWcv} GEMM, SGD.  ta3n_forward runs the first
// two, ta3n_loss nothing, ta3n_backward the third; ta3n_train_step all three.
int build_plan_avgpool(ta3n_plan &p, std::string &err) {
    const ta3n_config &c = p.cfg;
    const int Bs = c.batch_source, Bt = c.batch_target, T = c.num_segments, D = c.feature_dim;
    const int F = std::min(c.fc_dim, c.feature_dim), C = c.num_class;
    const int B = Bs + Bt, BT = B * T;
    const uint32_t da_flags = TA3N_FLAG_ADV_RELATION | TA3N_FLAG_ADV_VIDEO | TA3N_FLAG_ADV_FRAME | TA3N_FLAG_ATTN_ENTROPY | TA3N_FLAG_TRANS_ATTN;
    if (c.flags & da_flags) { err = "avgpool is built for the source-only configuration: no adversarial / attention flags (SURVEY 8f rank 4)"; return TA3N_ERR_INVALID; }
    if (T < 1 || T > 64) { err = "num_segments must be in [1,64]"; return TA3N_ERR_INVALID; }
    if (F % 4 != 0) { err = "avgpool: fc_dim must be a multiple of 4"; return TA3N_ERR_INVALID; }
    p.n_tuples = 0;
    p.tuple_first.assign(1, 0);
    Builder b(p);
    // live parameters first (the all-reduce / optimiser operand), then the rest of the reference's state_dict for this configuration
    b.add_linear("fc_feature_shared_source", F, D, true);              // models.py:141
    p.first_floats = p.param_floats;
    b.add_linear("fc_classifier_video_source", C, F, true);            // :272 (feat_aggregated_dim = F, :246-247)
    p.live_floats = p.param_floats;
    b.add_linear("fc_feature_source", F, F, false);
    b.add_linear("fc_feature_domain", F, F, false);
    b.add_linear("fc_classifier_source", C, F, false);
    b.add_linear("fc_classifier_domain", 2, F, false);
    b.add_linear("fc_feature_video_source", F, F, false);
    b.add_linear("fc_feature_video_source_2", F, F, false);
    b.add_linear("fc_feature_domain_video", F, F, false);
    b.add_linear("fc_classifier_domain_video", 2, F, false);
    const int64_t Wsh = p.poff("fc_feature_shared_source.weight"), bsh = p.poff("fc_feature_shared_source.bias");
    const int64_t Wcv = p.poff("fc_classifier_video_source.weight"), bcv = p.poff("fc_classifier_video_source.bias");

    Geom &g = p.geom;
    std::memset(&g, 0, sizeof(g));
    g.Bs = Bs; g.Bt = Bt; g.B = B; g.T = T; g.D = D; g.F = F; g.NB = F; g.C = C;
    g.flags = c.flags;
    g.o_ws16 = g.o_p16 = g.o_x16 = g.o_p16b = -1; g.pair_delta = 0;
    g.o_F1 = (int32_t)b.add_region("F1", (int64_t)BT * F);
    g.o_V = (int32_t)b.add_region("V", (int64_t)B * F);
    g.o_Vd = (int32_t)b.add_region("Vd", (int64_t)B * F);
    g.o_Y = (int32_t)b.add_region("Y", (int64_t)B * C);
    g.o_gY = (int32_t)b.add_region("gY", (int64_t)B * C);
    g.o_gZ1 = (int32_t)b.add_region("gZ1", (int64_t)BT * F);
    g.o_zeros = (int32_t)b.add_region("zeros", 64);
    g.o_ones = (int32_t)b.add_region("ones", (int64_t)BT * 4);
    if (g.o_ones != g.o_zeros + 64) { err = "internal: ones must follow zeros"; return TA3N_ERR_INVALID; }
    g.o_losses = (int32_t)b.add_region("losses", 8);
    g.n_norm_blocks = 256;
    g.o_norm_part = (int32_t)b.add_region("norm_part", g.n_norm_blocks);
    g.o_grad_norm = (int32_t)b.add_region("grad_norm", 4);
    g.o_hyper = (int32_t)b.add_region("hyper", 32);
    g.o_labels = (int32_t)b.add_region("labels", B);
    g.o_tuple_first = (int32_t)b.add_region("tuple_first", 1);
    g.n_vid_wg = B; g.n_frm_wg = 0;
    g.o_loss_part = (int32_t)b.add_region("loss_part", (int64_t)B * 8);
    g.o_metrics = (int32_t)b.add_region("metrics", 8);
    g.o_confusion = (int32_t)b.add_region("confusion", (int64_t)C * C);
    g.live_floats = (int32_t)p.live_floats;
    g.p_Wcv = (int32_t)Wcv; g.p_bcv = (int32_t)bcv;

    int split_shared_fc = 0;    // (set for the fused step's first launch: ta3n_config.split_k bit 2)
    auto spec_F1 = [&]() {   // shared frame FC + ReLU + dropout_i (models.py:565-575)
        GemmSpec s;
        s.M = BT; s.N = F;
        if (split_shared_fc && D >= 512 && (D / 2) % 128 == 0) {
            // split_k bit 2 (round 6 experiment): the launch has ONE tile per compute unit at the headline shape - one resident workgroup
            // pulls its stages at ~15 B/clk where two pull ~21 - so every tile becomes two workgroups over the two halves of K
            // (EPI_SPLITK: partial tile through L2 + ticket, the second to arrive adds and runs the epilogue)
            s.segs.push_back(mkseg(KC(BASE_X, 0, D), KC(BASE_P, Wsh, D), D / 2));
            s.segs.push_back(mkseg(KC(BASE_X, D / 2, D), KC(BASE_P, Wsh + D / 2, D), D / 2));
            s.split = 2;
        } else
        s.segs.push_back(mkseg(KC(BASE_X, 0, D), KC(BASE_P, Wsh, D), D));
        s.proto = proto(BASE_WS, g.o_F1, F);
        with_bias(s.proto, bsh);
        s.proto.epi |= EPI_RELU | EPI_DROP_I;
        s.proto.gamma_kind = SK_INV_KEEP_I;
        s.proto.drop_ld = F;
        return s;
    };
    auto grads = [&]() {
        std::vector<GemmSpec> s;
        GemmSpec gw;             // dWsh = gZ1^T X, dbsh = column sums of gZ1
        gw.M = F; gw.N = D;
        gw.segs.push_back(mkseg(KM(BASE_WS, g.o_gZ1, F), KM(BASE_X, 0, D), BT));
        gw.proto = proto(BASE_G, Wsh, D);
        gw.proto.epi |= EPI_ROWSUM_A; gw.proto.bias_base = BASE_G; gw.proto.bias_off = (int32_t)bsh;
        s.push_back(gw);
        GemmSpec gc;             // dWcv = gY^T Vd, dbcv = column sums of gY
        gc.M = C; gc.N = F;
        gc.segs.push_back(mkseg(KM(BASE_WS, g.o_gY, C), KM(BASE_WS, g.o_Vd, F), B));
        gc.proto = proto(BASE_G, Wcv, F);
        gc.proto.epi |= EPI_ROWSUM_A; gc.proto.bias_base = BASE_G; gc.proto.bias_off = (int32_t)bcv;
        s.push_back(gc);
        return s;
    };
    for (int group : {0, 4}) {           // ta3n_forward / first part of ta3n_train_step
        std::vector<GemmSpec> s{spec_F1()};
        b.add_gemm_phase(group, s);
        b.add_simple_phase(PH_POOL_CLS, group);
        if (group == 4) b.sum8_pending(g.o_losses, g.o_loss_part, B);
        if (group == 4) { auto q = grads(); b.add_gemm_phase(4, q); }
    }
    b.sum8_pending(g.o_losses, g.o_loss_part, B);
    { auto q = grads(); b.add_gemm_phase(2, q); }   // ta3n_backward (also adds up the loss partials for logging)
    b.add_simple_phase(PH_GRAD_NORM, 3);
    b.add_simple_phase(PH_SGD, 3);
    // fused grad-norm partials, as in the trn-m step
    std::vector<size_t> grad_tasks;
    for (const Phase &ph : p.phases)
        if (ph.group == 4 && ph.kind == PH_GEMM)
            for (int i = ph.task_begin; i < ph.task_begin + ph.task_count; ++i)
                if ((p.tasks[i].seg_count > 0 || (p.tasks[i].epi & EPI_COLSUM)) && p.tasks[i].c_base == BASE_G) grad_tasks.push_back((size_t)i);
    g.n_sumsq = (int32_t)grad_tasks.size();
    g.o_sumsq = (int32_t)b.add_region("sumsq", g.n_sumsq);
    for (size_t k = 0; k < grad_tasks.size(); ++k) {
        p.tasks[grad_tasks[k]].epi |= EPI_SUMSQ;
        p.tasks[grad_tasks[k]].pad[3] = g.o_sumsq + (int32_t)k;
    }
    add_bf16_twins(p, b, g, BT, D, {Span{g.o_gZ1, g.o_gZ1 + (int64_t)BT * F}});   // pool_cls keeps the twin of gZ1
    if (p.ws_floats >= (1ll << 31)) { err = "workspace too large for 32-bit offsets"; return TA3N_ERR_INVALID; }
    if (b.mixed_kinds) { err = "internal: a GEMM spec mixes operand kinds across its K segments"; return TA3N_ERR_INVALID; }
    for (auto &t : p.tasks)      // (after the twin re-addressing: the copies must be the final Segs)
        if (t.seg_count > 0) t.seg0 = p.segs[t.seg_begin];
    return TA3N_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// TA3N_AGG_AVGPOOL, general: TemPooling with the adversarial branches (the "RevGrad" rows of the paper's TemPooling table;
// also what the module path uses for every avgpool model, because there the caller assembles the loss).  Reference:
// models.py:557-579 (shared frame FC), 456-462 (frame discriminator), 421-433 (averaging), 679-687 (dropout_v, classifier),
// 464-470 (video discriminator; feat_aggregated_dim = F, :246-247), 697-708 (pred_domain = [video again, video, frame]),
// main.py:439-451, 508-538 (losses), 574-583.  Every Linear is a tile-list GEMM; the three small kernels are the
// averaging, the losses (the trn-m loss kernel with zero relation rows) and the way back through the averaging.
// Launches of one step (ta3n_forward / ta3n_loss / ta3n_backward, or all of them as ta3n_train_step): F1 | Hf | mean |
// {Y, Hv} | {Pv, Pf} | losses | {gHv, gHf, small weight gradients} | {gVt, dWdv, dWfd} | spread | gZ1 | dWsh.
int build_plan_avgpool_general(ta3n_plan &p, std::string &err) {
    const ta3n_config &c = p.cfg;
    const int Bs = c.batch_source, Bt = c.batch_target, T = c.num_segments, D = c.feature_dim;
    const int F = std::min(c.fc_dim, c.feature_dim), C = c.num_class;
    const int B = Bs + Bt, BT = B * T;
    if (c.flags & (TA3N_FLAG_ATTN_ENTROPY | TA3N_FLAG_TRANS_ATTN)) {
        err = "avgpool: attention / attentive entropy are not built (the reference's script switches them off with use_attn none)";
        return TA3N_ERR_INVALID;
    }
    if (T < 1 || T > 64) { err = "num_segments must be in [1,64]"; return TA3N_ERR_INVALID; }
    if (F % 4 != 0) { err = "avgpool: fc_dim must be a multiple of 4"; return TA3N_ERR_INVALID; }
    const bool live_frm = (c.flags & TA3N_FLAG_ADV_FRAME) != 0;
    // the video discriminator also serves the relation slot (models.py:707-708), see loss_kernel
    const bool live_vid = (c.flags & (TA3N_FLAG_ADV_VIDEO | TA3N_FLAG_ADV_RELATION)) != 0;
    p.n_tuples = 0;
    p.tuple_first.assign(1, 0);
    Builder b(p);
    struct Lin { std::string name; int out, in; };
    std::vector<Lin> dead;
    auto lin = [&](const std::string &name, int out, int in, bool live) {
        if (live) b.add_linear(name, out, in, true);
        else dead.push_back(Lin{name, out, in});
    };
    b.add_linear("fc_feature_shared_source", F, D, true);              // models.py:141
    p.first_floats = p.param_floats;
    lin("fc_feature_domain", F, F, live_frm);                          // :161
    lin("fc_classifier_domain", 2, F, live_frm);                       // :170
    lin("fc_feature_domain_video", F, F, live_vid);                    // :267 (feat_aggregated_dim = F)
    b.add_linear("fc_classifier_video_source", C, F, true);            // :272
    const bool mcd = (c.flags & TA3N_FLAG_MCD) != 0;                   // the DA options of the TemPooling rows (module path)
    const bool feat_grads = (c.flags & TA3N_FLAG_FEATURE_GRADS) != 0;
    const bool bn_shared = (c.flags & TA3N_FLAG_BN_SHARED) != 0;
    if (bn_shared && F % BN_COLS != 0) { err = "use_bn: fc_dim must be a multiple of 4 (the BatchNorm launches move a row's four columns as one 16-byte access)"; return TA3N_ERR_INVALID; }
    if (mcd) b.add_linear("fc_classifier_video_source_2", C, F, true); // :276-279
    if (bn_shared) {                                                   // :195-196
        b.add_param("bn_shared_S.weight", F, 0, true); b.add_param("bn_shared_S.bias", F, 0, true);
        b.add_param("bn_shared_T.weight", F, 0, true); b.add_param("bn_shared_T.bias", F, 0, true);
    }
    lin("fc_classifier_domain_video", 2, F, live_vid);                 // :281
    p.live_floats = p.param_floats;
    for (auto &d : dead) b.add_linear(d.name, d.out, d.in, false);
    b.add_linear("fc_feature_source", F, F, false);
    b.add_linear("fc_classifier_source", C, F, false);
    b.add_linear("fc_feature_video_source", F, F, false);
    b.add_linear("fc_feature_video_source_2", F, F, false);
    auto P = [&](const std::string &n) { return p.poff(n); };
    const int64_t Wsh = P("fc_feature_shared_source.weight"), bsh = P("fc_feature_shared_source.bias");
    const int64_t Wfd = P("fc_feature_domain.weight"), bfd = P("fc_feature_domain.bias");
    const int64_t Wcd = P("fc_classifier_domain.weight"), bcd = P("fc_classifier_domain.bias");
    const int64_t Wdv = P("fc_feature_domain_video.weight"), bdv = P("fc_feature_domain_video.bias");
    const int64_t Wcv = P("fc_classifier_video_source.weight"), bcv = P("fc_classifier_video_source.bias");
    const int64_t Wcdv = P("fc_classifier_domain_video.weight"), bcdv = P("fc_classifier_domain_video.bias");

    Geom &g = p.geom;
    std::memset(&g, 0, sizeof(g));
    g.Bs = Bs; g.Bt = Bt; g.B = B; g.T = T; g.D = D; g.F = F; g.NB = F; g.C = C;
    g.n_tuples = 0; g.n_rel = 0; g.flags = c.flags;
    g.o_ws16 = g.o_p16 = g.o_x16 = g.o_p16b = -1; g.pair_delta = 0;
    g.o_F1 = (int32_t)b.add_region("F1", (int64_t)BT * F);
    g.o_Hf = (int32_t)b.add_region("Hf", live_frm ? (int64_t)BT * F : 4);
    g.o_Pf = (int32_t)b.add_region("Pf", (int64_t)BT * 2);
    g.o_V = (int32_t)b.add_region("V", (int64_t)B * F);
    g.o_Vd = (int32_t)b.add_region("Vd", (int64_t)B * F);
    g.o_Y = (int32_t)b.add_region("Y", (int64_t)B * C);
    g.o_Hv = (int32_t)b.add_region("Hv", live_vid ? (int64_t)B * F : 4);
    g.o_Pv = (int32_t)b.add_region("Pv", (int64_t)B * 2);
    g.o_Pr = (int32_t)b.add_region("Pr", 4);            // no relation rows
    g.o_gY = (int32_t)b.add_region("gY", (int64_t)B * C);
    g.o_gPv = (int32_t)b.add_region("gPv", (int64_t)B * 2);
    g.o_gPr = (int32_t)b.add_region("gPr", 4);
    g.o_gPf = (int32_t)b.add_region("gPf", (int64_t)BT * 2);
    g.o_gHv = (int32_t)b.add_region("gHv", live_vid ? (int64_t)B * F : 4);
    const int64_t o_gHf = b.add_region("gHf", live_frm ? (int64_t)BT * F : 4);
    g.o_gHf = live_frm ? (int32_t)o_gHf : -1;           // -1: pool_avg_bwd writes gZ1 directly
    g.o_gVt = (int32_t)b.add_region("gVt", (int64_t)B * F);
    g.o_gRa = (int32_t)b.add_region("gRa", live_frm ? (int64_t)BT * F : 4);   // gVt / T spread over the segments
    g.o_gZ1 = (int32_t)b.add_region("gZ1", (int64_t)BT * F);
    g.o_attn = (int32_t)b.add_region("attn", 4); g.o_gattn = (int32_t)b.add_region("g_attn", 4);
    const int64_t Wcv2 = mcd ? P("fc_classifier_video_source_2.weight") : 0, bcv2 = mcd ? P("fc_classifier_video_source_2.bias") : 0;
    if (feat_grads) g.o_gV_ext = (int32_t)b.add_region("gV_ext", (int64_t)B * F);
    if (mcd) {
        g.o_Y2 = (int32_t)b.add_region("Y2", (int64_t)B * C);
        g.o_gY2 = (int32_t)b.add_region("gY2", (int64_t)B * C);
    }
    if (bn_shared) {
        g.o_Z0 = (int32_t)b.add_region("Z0", (int64_t)BT * F);
        g.o_gZ0 = (int32_t)b.add_region("gZ0", (int64_t)BT * F);
        g.o_bn_batch = (int32_t)b.add_region("bn_batch", (int64_t)2 * 3 * F);
        g.o_bn_run = (int32_t)b.add_region("bn_run", (int64_t)2 * 2 * F);
        g.p_bn_w[0] = (int32_t)P("bn_shared_S.weight"); g.p_bn_b[0] = (int32_t)P("bn_shared_S.bias");
        g.p_bn_w[1] = (int32_t)P("bn_shared_T.weight"); g.p_bn_b[1] = (int32_t)P("bn_shared_T.bias");
    }
    g.o_zeros = (int32_t)b.add_region("zeros", 64);
    g.o_ones = (int32_t)b.add_region("ones", (int64_t)BT * 4);
    if (g.o_ones != g.o_zeros + 64) { err = "internal: ones must follow zeros"; return TA3N_ERR_INVALID; }
    g.o_losses = (int32_t)b.add_region("losses", 8);
    g.n_norm_blocks = 256;
    g.o_norm_part = (int32_t)b.add_region("norm_part", g.n_norm_blocks);
    g.o_grad_norm = (int32_t)b.add_region("grad_norm", 4);
    g.o_hyper = (int32_t)b.add_region("hyper", 32);
    g.o_labels = (int32_t)b.add_region("labels", B);
    g.o_tuple_first = (int32_t)b.add_region("tuple_first", 1);
    g.o_loss_part = (int32_t)b.add_region("loss_part", 8);
    g.o_metrics = (int32_t)b.add_region("metrics", 8);
    g.o_confusion = (int32_t)b.add_region("confusion", (int64_t)C * C);
    g.live_floats = (int32_t)p.live_floats;
    g.p_Wcv = (int32_t)Wcv; g.p_bcv = (int32_t)bcv;

    auto fwd = [&](int M, int N, int K, int64_t x_off, int64_t w, int64_t bias, int64_t y_off, bool relu) {   // Y = act(X W^T + b)
        GemmSpec s;
        s.M = M; s.N = N;
        s.segs.push_back(mkseg(KC(BASE_WS, x_off, K), KC(BASE_P, w, K), K));
        s.proto = proto(BASE_WS, y_off, N);
        with_bias(s.proto, bias);
        if (relu) s.proto.epi |= EPI_RELU;
        return s;
    };
    int split_shared_fc = 0;    // (set for the fused step's first launch: ta3n_config.split_k bit 2)
    auto spec_F1 = [&]() {   // shared frame FC + ReLU + dropout_i (models.py:565-575)
        GemmSpec s;
        s.M = BT; s.N = F;
        if (split_shared_fc && D >= 512 && (D / 2) % 128 == 0) {
            // split_k bit 2 (round 6 experiment): the launch has ONE tile per compute unit at the headline shape - one resident workgroup
            // pulls its stages at ~15 B/clk where two pull ~21 - so every tile becomes two workgroups over the two halves of K
            // (EPI_SPLITK: partial tile through L2 + ticket, the second to arrive adds and runs the epilogue)
            s.segs.push_back(mkseg(KC(BASE_X, 0, D), KC(BASE_P, Wsh, D), D / 2));
            s.segs.push_back(mkseg(KC(BASE_X, D / 2, D), KC(BASE_P, Wsh + D / 2, D), D / 2));
            s.split = 2;
        } else
        s.segs.push_back(mkseg(KC(BASE_X, 0, D), KC(BASE_P, Wsh, D), D));
        if (bn_shared) {   // the linear output only: BatchNorm, ReLU and dropout follow in the PH_BN_FWD launch
            s.proto = proto(BASE_WS, g.o_Z0, F);
            with_bias(s.proto, bsh);
            return s;
        }
        s.proto = proto(BASE_WS, g.o_F1, F);
        with_bias(s.proto, bsh);
        s.proto.epi |= EPI_RELU | EPI_DROP_I;
        s.proto.gamma_kind = SK_INV_KEEP_I;
        s.proto.drop_ld = F;
        return s;
    };
    auto masked_back = [&](int M, int N, int64_t g_off, int64_t w, int64_t mask_off, int64_t out_off) {   // (G W) * [H > 0], G two logits wide
        GemmSpec s;
        s.M = M; s.N = N;
        s.segs.push_back(mkseg(KC(BASE_WS, g_off, 2), KM(BASE_P, w, N), 2));
        s.proto = proto(BASE_WS, out_off, N);
        with_mask(s.proto, mask_off, N);
        return s;
    };
    auto wgrad = [&](int M, int N, int K, int64_t g_off, int g_ld, int base_x, int64_t x_off, int x_ld, int64_t dst, int64_t bias_dst) {
        GemmSpec gw;
        gw.M = M; gw.N = N;
        gw.segs.push_back(mkseg(KM(BASE_WS, g_off, g_ld), KM(base_x, x_off, x_ld), K));
        gw.proto = proto(BASE_G, dst, N);
        gw.proto.epi |= EPI_ROWSUM_A; gw.proto.bias_base = BASE_G; gw.proto.bias_off = (int32_t)bias_dst;
        return gw;
    };
    auto forward = [&](int group) {
        { std::vector<GemmSpec> s{spec_F1()}; b.add_gemm_phase(group, s); }
        if (bn_shared) b.add_simple_phase(PH_BN_FWD, group);
        if (live_frm) { std::vector<GemmSpec> s{fwd(BT, F, F, g.o_F1, Wfd, bfd, g.o_Hf, true)}; b.add_gemm_phase(group, s); }   // models.py:458-459
        b.add_simple_phase(PH_POOL_AVG_FWD, group);
        {
            std::vector<GemmSpec> s{fwd(B, C, F, g.o_Vd, Wcv, bcv, g.o_Y, false)};                                 // :686
            if (mcd) s.push_back(fwd(B, C, F, g.o_Vd, Wcv2, bcv2, g.o_Y2, false));                                  // :717-718
            if (live_vid) s.push_back(fwd(B, F, F, g.o_Vd, Wdv, bdv, g.o_Hv, true));                                // :466-467
            b.add_gemm_phase(group, s);
        }
        if (live_vid || live_frm) {
            std::vector<GemmSpec> s;
            if (live_vid) s.push_back(fwd(B, 2, F, g.o_Hv, Wcdv, bcdv, g.o_Pv, false));                             // :468
            if (live_frm) s.push_back(fwd(BT, 2, F, g.o_Hf, Wcd, bcd, g.o_Pf, false));                              // :460
            b.add_gemm_phase(group, s);
        }
    };
    auto backward = [&](int group) {
        {   // what depends only on the logit gradients
            std::vector<GemmSpec> s;
            if (live_vid) {
                s.push_back(masked_back(B, F, g.o_gPv, Wcdv, g.o_Hv, g.o_gHv));
                s.push_back(wgrad(2, F, B, g.o_gPv, 2, BASE_WS, g.o_Hv, F, Wcdv, bcdv));
            }
            if (live_frm) {
                s.push_back(masked_back(BT, F, g.o_gPf, Wcd, g.o_Hf, g.o_gHf));
                s.push_back(wgrad(2, F, BT, g.o_gPf, 2, BASE_WS, g.o_Hf, F, Wcd, bcd));
            }
            s.push_back(wgrad(C, F, B, g.o_gY, C, BASE_WS, g.o_Vd, F, Wcv, bcv));
            if (mcd) s.push_back(wgrad(C, F, B, g.o_gY2, C, BASE_WS, g.o_Vd, F, Wcv2, bcv2));
            b.add_gemm_phase(group, s);
        }
        {   // gVt = dropout_v'( -beta1 * gHv Wdv + gY Wcv ), first-layer weight gradients of the discriminators
            std::vector<GemmSpec> s;
            GemmSpec gv;
            gv.M = B; gv.N = F;
            if (live_vid) gv.segs.push_back(mkseg(KC(BASE_WS, g.o_gHv, F), KM(BASE_P, Wdv, F), F, SK_NEG_BETA_VID));
            gv.segs.push_back(mkseg(KC(BASE_WS, g.o_gY, C), KM(BASE_P, Wcv, F), C));
            if (mcd) gv.segs.push_back(mkseg(KC(BASE_WS, g.o_gY2, C), KM(BASE_P, Wcv2, F), C));
            gv.proto = proto(BASE_WS, g.o_gVt, F);
            gv.proto.alpha_kind = SK_REVERSE_MU;      // forward(..., reverse=True): GradReverse(mu) behind dropout_v (models.py:682-684)
            gv.proto.epi |= EPI_DROP_V; gv.proto.gamma_kind = SK_INV_KEEP_V; gv.proto.drop_ld = F;
            s.push_back(gv);
            if (live_vid) s.push_back(wgrad(F, F, B, g.o_gHv, F, BASE_WS, g.o_Vd, F, Wdv, bdv));
            if (live_frm) s.push_back(wgrad(F, F, BT, g.o_gHf, F, BASE_WS, g.o_F1, F, Wfd, bfd));
            b.add_gemm_phase(group, s);
        }
        b.add_simple_phase(PH_POOL_AVG_BWD, group);
        if (live_frm) {   // gZ1 = ( -beta2 gHf Wfd + gVt / T ) * [F1 > 0] / keep_i
            GemmSpec gz;
            gz.M = BT; gz.N = F;
            gz.segs.push_back(mkseg(KC(BASE_WS, g.o_gHf, F), KM(BASE_P, Wfd, F), F, SK_NEG_BETA_FRM));
            gz.proto = proto(BASE_WS, g.o_gZ1, F);
            with_add(gz.proto, g.o_gRa, F);
            with_mask(gz.proto, g.o_F1, F);
            gz.proto.gamma_kind = SK_INV_KEEP_I;
            std::vector<GemmSpec> s{gz};
            b.add_gemm_phase(group, s);
        }
        if (bn_shared) b.add_simple_phase(PH_BN_BWD, group);
        { std::vector<GemmSpec> s{wgrad(F, D, BT, bn_shared ? g.o_gZ0 : g.o_gZ1, F, BASE_X, 0, D, Wsh, bsh)}; b.add_gemm_phase(group, s); }
    };
    forward(0);
    b.add_simple_phase(PH_LOSS, 1);
    backward(2);
    b.add_simple_phase(PH_GRAD_NORM, 3);
    b.add_simple_phase(PH_SGD, 3);
    forward(4);                           // ta3n_train_step: the same launches as one sequence
    b.add_simple_phase(PH_LOSS, 4);
    backward(4);
    std::vector<size_t> grad_tasks;
    for (const Phase &ph : p.phases)
        if (ph.group == 4 && ph.kind == PH_GEMM)
            for (int i = ph.task_begin; i < ph.task_begin + ph.task_count; ++i)
                if (p.tasks[i].seg_count > 0 && p.tasks[i].c_base == BASE_G) grad_tasks.push_back((size_t)i);
    // use_bn: the BatchNorm weight / bias gradients come from the PH_BN_BWD launch - its workgroups leave their sums of squares in the LAST
    // 2 * ((F + BN_COLS - 1) / BN_COLS) slots of the region (bn_shared_bwd_kernel), as in the trn-m plan
    g.n_sumsq = (int32_t)grad_tasks.size() + (bn_shared ? 2 * ((F + BN_COLS - 1) / BN_COLS) : 0);
    g.o_sumsq = (int32_t)b.add_region("sumsq", g.n_sumsq);
    for (size_t k = 0; k < grad_tasks.size(); ++k) {
        p.tasks[grad_tasks[k]].epi |= EPI_SUMSQ;
        p.tasks[grad_tasks[k]].pad[3] = g.o_sumsq + (int32_t)k;
    }
    // (no bf16 twins on this path: the small kernels between the launches do not maintain them; the contractions still
    // round their operands in registers with TA3N_FLAG_BF16_MFMA)
    if (p.ws_floats >= (1ll << 31)) { err = "workspace too large for 32-bit offsets"; return TA3N_ERR_INVALID; }
    if (b.mixed_kinds) { err = "internal: a GEMM spec mixes operand kinds across its K segments"; return TA3N_ERR_INVALID; }
    for (auto &t : p.tasks)
        if (t.seg_count > 0) t.seg0 = p.segs[t.seg_begin];
    return TA3N_OK;
}

}  // namespace

static int build_plan_once(ta3n_plan &p, std::string &err) {
    const ta3n_config &c = p.cfg;
    const int Bs = c.batch_source, Bt = c.batch_target, T = c.num_segments, D = c.feature_dim;
    const int F = std::min(c.fc_dim, c.feature_dim);   // models.py:129
    const int NB = c.num_bottleneck, C = c.num_class;
    if (Bs < 0 || Bt < 0 || Bs + Bt <= 0) { err = "batch sizes must be non-negative and not both zero"; return TA3N_ERR_INVALID; }
    if (c.aggregation != TA3N_AGG_TRN_M && c.aggregation != TA3N_AGG_AVGPOOL) { err = "aggregation must be TA3N_AGG_TRN_M or TA3N_AGG_AVGPOOL"; return TA3N_ERR_INVALID; }
    if (T < 2 || T > 64) { err = "num_segments must be in [2,64] for trn-m"; return TA3N_ERR_INVALID; }
    if (D <= 0 || F <= 0 || C <= 0) { err = "feature_dim, fc_dim and num_class must be positive"; return TA3N_ERR_INVALID; }
    if (NB <= 0 || NB % 64 != 0 || NB > 1024) { err = "num_bottleneck must be a multiple of 64 (<= 1024)"; return TA3N_ERR_INVALID; }
    if (C > 64) { err = "num_class > 64 not supported by the loss kernel"; return TA3N_ERR_INVALID; }
    if ((c.flags & TA3N_FLAG_ATTN_ENTROPY) &&
        !((c.flags & TA3N_FLAG_ADV_RELATION) && (c.flags & TA3N_FLAG_ADV_VIDEO))) {
        // main.py:559-562 indexes pred_domain_all[1]; that is the video entry only when
        // place_adv[0] and place_adv[1] are both 'Y' (otherwise the reference fails on a shape mismatch)
        err = "attentive_entropy requires place_adv[0]=='Y' and place_adv[1]=='Y'";
        return TA3N_ERR_INVALID;
    }
    if ((c.flags & TA3N_FLAG_F32_SPLIT) && (c.flags & TA3N_FLAG_BF16_MFMA)) {
        err = "TA3N_FLAG_F32_SPLIT and TA3N_FLAG_BF16_MFMA are different arithmetics: set one";
        return TA3N_ERR_INVALID;
    }
    auto tile_ok = [](int t) { return t == 0 || tile_config_ok(t); };
    if (!tile_ok(c.tile_config)) { err = "tile_config must be 0 or one of 114, 118, 212, 122, 214, 124, 221, 222 (+ 2000 / 3000: bf16 stages; + 10000 / 20000 / 30000: 2 row / 2 column / 2 x 2 blocks per wave with 222 or 221, bf16 twins)"; return TA3N_ERR_INVALID; }
    for (int i = 0; i < 16; ++i)
        if (!tile_ok(c.phase_tiles[i])) { err = "phase_tiles entries must be 0 or one of 114, 118, 212, 122, 214, 124, 221, 222 (+ 2000 / 3000: bf16 stages; + 10000 / 20000 / 30000: 2 row / 2 column / 2 x 2 blocks per wave with 222 or 221, bf16 twins)"; return TA3N_ERR_INVALID; }
    if (c.xcd_aware < 0 || c.xcd_aware > 3) { err = "xcd_aware must be 0, 1, 2 or 3"; return TA3N_ERR_INVALID; }
    const int B = Bs + Bt, BT = B * T, NR = T - 1;
    if ((int64_t)BT * D >= (1ll << 31) || (int64_t)BT * F >= (1ll << 31)) { err = "problem too large for 32-bit offsets"; return TA3N_ERR_INVALID; }
    if (c.aggregation == TA3N_AGG_AVGPOOL)      // source-only: the fused fast path (BASELINE configs[0]); with adversarial branches or a module-path option: the general one
        return (c.flags & (TA3N_FLAG_ADV_RELATION | TA3N_FLAG_ADV_VIDEO | TA3N_FLAG_ADV_FRAME | TA3N_FLAG_MCD | TA3N_FLAG_FEATURE_GRADS |
                           TA3N_FLAG_BN_SHARED)) ? build_plan_avgpool_general(p, err)
                                                                                                 : build_plan_avgpool(p, err);

    // ---- relation tuples ----
    p.n_tuples = ta3n_num_relation_tuples(T);
    p.tuples.assign((size_t)p.n_tuples * T, -1);
    p.scale_len.assign(p.n_tuples, 0);
    p.scale_id.assign(p.n_tuples, 0);
    ta3n_relation_table(T, p.tuples.data(), p.scale_len.data(), p.scale_id.data());
    const int NT = p.n_tuples;
    p.tuple_first.assign(NR + 1, 0);
    for (int t = 0; t < NT; ++t) p.tuple_first[p.scale_id[t] + 1] = t + 1;
    for (int j = 1; j <= NR; ++j) p.tuple_first[j] = std::max(p.tuple_first[j], p.tuple_first[j - 1]);

    Builder b(p);
    // ---- parameters: live first (they form the all-reduce / optimiser operand) ----
    // Which discriminators receive a gradient depends on the options, as in the reference, where a parameter whose
    // output feeds no loss keeps grad None and is skipped by SGD (no weight decay either): the frame discriminator is
    // live only with its adversarial loss (place_adv[2], main.py:508-538), the video discriminator with its loss
    // (attentive entropy needs it too, and the plan requires ADV_VIDEO for it), the relation discriminators with their
    // loss OR the transferable attention, whose weights are not detached (models.py:351-357, 379-388).
    const bool live_frm = (c.flags & TA3N_FLAG_ADV_FRAME) != 0;
    const bool live_vid = (c.flags & TA3N_FLAG_ADV_VIDEO) != 0;
    const bool live_rel = (c.flags & (TA3N_FLAG_ADV_RELATION | TA3N_FLAG_TRANS_ATTN)) != 0;
    struct Lin { std::string name; int out, in; };
    std::vector<Lin> dead;
    auto lin = [&](const std::string &name, int out, int in, bool live) {
        if (live) b.add_linear(name, out, in, true);
        else dead.push_back(Lin{name, out, in});
    };
    b.add_linear("fc_feature_shared_source", F, D, true);              // models.py:141 (first: ta3n_train_step_after_update relies on it)
    p.first_floats = p.param_floats;
    lin("fc_feature_domain", F, F, live_frm);                          // :161
    lin("fc_classifier_domain", 2, F, live_frm);                       // :170
    for (int j = 0; j < NR; ++j)                                       // TRNmodule.py:44-54
        b.add_linear("TRN.fc_fusion_scales." + std::to_string(j) + ".1", NB, (T - j) * F, true);
    for (int j = 0; j < NR; ++j) {                                     // models.py:286-294
        lin("relation_domain_classifier_all." + std::to_string(j) + ".0", NB, NB, live_rel);
        lin("relation_domain_classifier_all." + std::to_string(j) + ".2", 2, NB, live_rel);
    }
    lin("fc_feature_domain_video", NB, NB, live_vid);                  // :267
    b.add_linear("fc_classifier_video_source", C, NB, true);           // :272
    const bool mcd = (c.flags & TA3N_FLAG_MCD) != 0;
    const bool feat_grads = (c.flags & TA3N_FLAG_FEATURE_GRADS) != 0;
    const bool bn_shared = (c.flags & TA3N_FLAG_BN_SHARED) != 0;
    if (bn_shared && F % BN_COLS != 0) { err = "use_bn: fc_dim must be a multiple of 4 (the BatchNorm launches move a row's four columns as one 16-byte access)"; return TA3N_ERR_INVALID; }
    if (bn_shared) {   // models.py:195-196 (nn.BatchNorm1d(feat_shared_dim) x 2): weight and bias are trained
        b.add_param("bn_shared_S.weight", F, 0, true); b.add_param("bn_shared_S.bias", F, 0, true);
        b.add_param("bn_shared_T.weight", F, 0, true); b.add_param("bn_shared_T.bias", F, 0, true);
    }
    if (mcd) b.add_linear("fc_classifier_video_source_2", C, NB, true); // :276-279 (ens_DA MCD)
    lin("fc_classifier_domain_video", 2, NB, live_vid);                // :281
    p.live_floats = p.param_floats;
    for (auto &d : dead) b.add_linear(d.name, d.out, d.in, false);     // computed (zero or unused) gradients land past the live prefix
    // never receive a gradient in this configuration: kept for state_dict compatibility
    b.add_linear("fc_feature_source", F, F, false);                    // :156
    b.add_linear("fc_classifier_source", C, F, false);                 // :166 (dead for baseline_type 'video')
    b.add_param("bn_trn_S.weight", NB, 0, false); b.add_param("bn_trn_S.bias", NB, 0, false);   // :225-226
    b.add_param("bn_trn_T.weight", NB, 0, false); b.add_param("bn_trn_T.bias", NB, 0, false);
    b.add_linear("fc_feature_video_source", NB, NB, false);            // :258
    b.add_linear("fc_feature_video_source_2", NB, NB, false);          // :262

    auto P = [&](const std::string &n) { return p.poff(n); };
    auto trnW = [&](int j) { return P("TRN.fc_fusion_scales." + std::to_string(j) + ".1.weight"); };
    auto trnB = [&](int j) { return P("TRN.fc_fusion_scales." + std::to_string(j) + ".1.bias"); };
    auto W1 = [&](int j) { return P("relation_domain_classifier_all." + std::to_string(j) + ".0.weight"); };
    auto B1 = [&](int j) { return P("relation_domain_classifier_all." + std::to_string(j) + ".0.bias"); };
    auto W2 = [&](int j) { return P("relation_domain_classifier_all." + std::to_string(j) + ".2.weight"); };
    auto B2 = [&](int j) { return P("relation_domain_classifier_all." + std::to_string(j) + ".2.bias"); };
    const int64_t Wsh = P("fc_feature_shared_source.weight"), bsh = P("fc_feature_shared_source.bias");
    const int64_t Wfd = P("fc_feature_domain.weight"), bfd = P("fc_feature_domain.bias");
    const int64_t Wcd = P("fc_classifier_domain.weight"), bcd = P("fc_classifier_domain.bias");
    const int64_t Wdv = P("fc_feature_domain_video.weight"), bdv = P("fc_feature_domain_video.bias");
    const int64_t Wcv = P("fc_classifier_video_source.weight"), bcv = P("fc_classifier_video_source.bias");
    const int64_t Wcdv = P("fc_classifier_domain_video.weight"), bcdv = P("fc_classifier_domain_video.bias");
    const int64_t Wcv2 = mcd ? P("fc_classifier_video_source_2.weight") : 0, bcv2 = mcd ? P("fc_classifier_video_source_2.bias") : 0;

    // ---- workspace ----
    Geom &g = p.geom;
    std::memset(&g, 0, sizeof(g));
    g.Bs = Bs; g.Bt = Bt; g.B = B; g.T = T; g.D = D; g.F = F; g.NB = NB; g.C = C;
    g.n_tuples = NT; g.n_rel = NR; g.flags = c.flags;
    g.o_ws16 = g.o_p16 = g.o_x16 = g.o_p16b = -1; g.pair_delta = 0; g.ws16_span = 0;
    g.o_F1 = (int32_t)b.add_region("F1", (int64_t)BT * F);
    g.o_Hf = (int32_t)b.add_region("Hf", (int64_t)BT * F);
    g.o_Pf = (int32_t)b.add_region("Pf", (int64_t)BT * 2);
    g.o_Zr = (int32_t)b.add_region("Zr", (int64_t)B * NT * NB);
    g.o_Hr = (int32_t)b.add_region("Hr", (int64_t)B * NR * NB);
    g.o_Pr = (int32_t)b.add_region("Pr", (int64_t)B * NR * 2);
    g.o_R = (int32_t)b.add_region("R", (int64_t)B * NR * NB);
    g.o_attn = (int32_t)b.add_region("attn", (int64_t)B * NR);
    g.o_V = (int32_t)b.add_region("V", (int64_t)B * NB);
    g.o_Vd = (int32_t)b.add_region("Vd", (int64_t)B * NB);
    g.o_Y = (int32_t)b.add_region("Y", (int64_t)B * C);
    g.o_Hv = (int32_t)b.add_region("Hv", (int64_t)B * NB);
    g.o_Pv = (int32_t)b.add_region("Pv", (int64_t)B * 2);
    g.o_gY = (int32_t)b.add_region("gY", (int64_t)B * C);
    g.o_gPv = (int32_t)b.add_region("gPv", (int64_t)B * 2);
    g.o_gPr = (int32_t)b.add_region("gPr", (int64_t)B * NR * 2);
    g.o_gPf = (int32_t)b.add_region("gPf", (int64_t)BT * 2);
    g.o_gattn = (int32_t)b.add_region("g_attn", (int64_t)B * NR);
    g.o_gHv = (int32_t)b.add_region("gHv", (int64_t)B * NB);
    g.o_gHf = (int32_t)b.add_region("gHf", (int64_t)BT * F);
    g.o_gVt = (int32_t)b.add_region("gVt", (int64_t)B * NB);
    g.o_gPrT = (int32_t)b.add_region("gPrT", (int64_t)B * NR * 2);
    g.o_gRa = (int32_t)b.add_region("gRa", (int64_t)B * NR * NB);
    g.o_gHr = (int32_t)b.add_region("gHr", (int64_t)B * NR * NB);
    g.o_gR = (int32_t)b.add_region("gR", (int64_t)B * NR * NB);
    g.o_gZ = (int32_t)b.add_region("gZ", (int64_t)B * NT * NB);
    g.o_gZ1 = (int32_t)b.add_region("gZ1", (int64_t)BT * F);
    if (feat_grads) g.o_gV_ext = (int32_t)b.add_region("gV_ext", (int64_t)B * NB);
    if (bn_shared) {
        g.o_Z0 = (int32_t)b.add_region("Z0", (int64_t)BT * F);
        g.o_gZ0 = (int32_t)b.add_region("gZ0", (int64_t)BT * F);
        g.o_bn_batch = (int32_t)b.add_region("bn_batch", (int64_t)2 * 3 * F);
        g.o_bn_run = (int32_t)b.add_region("bn_run", (int64_t)2 * 2 * F);
        g.p_bn_w[0] = (int32_t)P("bn_shared_S.weight"); g.p_bn_b[0] = (int32_t)P("bn_shared_S.bias");
        g.p_bn_w[1] = (int32_t)P("bn_shared_T.weight"); g.p_bn_b[1] = (int32_t)P("bn_shared_T.bias");
    }
    if (mcd) {
        g.o_Y2 = (int32_t)b.add_region("Y2", (int64_t)B * C);
        g.o_gY2 = (int32_t)b.add_region("gY2", (int64_t)B * C);
    }
#ifdef TA3N_GEMM_STAMPS
    b.add_region("stamps", 8192 * 16 * 2);            // (debug build: per-workgroup cycle stamps of the GEMM launches, directly in front of "zeros")
#endif
    g.o_zeros = (int32_t)b.add_region("zeros", 64);   // never written: source of out-of-range operand elements
    g.o_ones = (int32_t)b.add_region("ones", (int64_t)BT * 4);   // [BT][4] block of ones (k-major A operand of the column sums)
    if (g.o_ones != g.o_zeros + 64) { err = "internal: ones must follow zeros"; return TA3N_ERR_INVALID; }
    g.o_losses = (int32_t)b.add_region("losses", 8);
    g.n_norm_blocks = 256;
    g.o_norm_part = (int32_t)b.add_region("norm_part", g.n_norm_blocks);
    g.o_grad_norm = (int32_t)b.add_region("grad_norm", 4);
    g.o_hyper = (int32_t)b.add_region("hyper", 32);
    g.o_labels = (int32_t)b.add_region("labels", B);
    g.o_tuple_first = (int32_t)b.add_region("tuple_first", NR + 1);
    // Videos per video workgroup (ta3n_heads.hip; TA3N_HEADS_VPW in the environment forces 1 / 2 for A/B runs).  A video workgroup owns
    // its compute unit, so what matters is how many ROUNDS the launch needs and how long a workgroup lives - and a workgroup's life is
    // dominated by the relation stages, whose per-wave chain grows with the relations a wave handles.  Measured (profiles/r04_heads_vpw_ab.txt):
    // 128+128 videos x 12 segments (configs[4]): 2 per workgroup turns 1.2 rounds into one, launch 40.6 -> 29.1 us, two-stream step 477 -> 470;
    // 512+512 x 9 (configs[3]): the video workgroups alone fill the chip whatever the packing - 4 per workgroup 79.5 -> 79.5 us on one box and
    // +40 us per step on another, 2 per workgroup -7 us on the launch and +6 on the step: left at one; 128+74 x 5 (headline): 2 per
    // workgroup 15.5 -> 20.1 us (every video has a compute unit already).
    // Round 5: with the relation loops pipelined (heads_kernel PIPE: a wave requests its next relation's operands while it reduces the current
    // one) a wave that handles twice the relations no longer takes twice as long: 512+512 x 9 on two videos per workgroup 81.0 -> 65.8 us on
    // the launch, 0.457 -> 0.441 ms on the step (two alternating repeats, profiles/r05_heads_vpw_tune.txt) - two per workgroup from 225 videos up.
    g.heads_vpw = B > 224 ? 2 : 1;
    // (A/B override: 1 or 2.  Four per workgroup - never a plan's choice - faulted on the round-5 build at 512+512 x 9 ("memory access fault",
    // gpurun_out of call 5) and is no longer accepted; the kernel template keeps the case.)
    if (const char *e = std::getenv("TA3N_HEADS_VPW")) { const int v = std::atoi(e); if (v == 1 || v == 2) g.heads_vpw = v; }
    g.n_vid_wg = (B + g.heads_vpw - 1) / g.heads_vpw;
    g.heads_rpw = HEADS_RPW;      // all workgroups of the heads kernel resident at once if the chip (256 CUs) can hold them
    while (g.n_vid_wg + (BT + g.heads_rpw - 1) / g.heads_rpw > 256 && g.heads_rpw < 8 * HEADS_RPW) g.heads_rpw *= 2;
    g.n_frm_wg = (BT + g.heads_rpw - 1) / g.heads_rpw;
    g.o_fh_part = (int32_t)b.add_region("fh_part", (int64_t)g.n_frm_wg * 2 * F);
    g.o_fh_bpart = (int32_t)b.add_region("fh_bpart", (int64_t)g.n_frm_wg * 2);
    g.o_loss_part = (int32_t)b.add_region("loss_part", (int64_t)(g.n_vid_wg + g.n_frm_wg) * 8);
    g.o_metrics = (int32_t)b.add_region("metrics", 8);
    g.o_confusion = (int32_t)b.add_region("confusion", (int64_t)C * C);
    g.live_floats = (int32_t)p.live_floats;
    g.p_Wcd = (int32_t)Wcd; g.p_bcd = (int32_t)bcd; g.p_Wdv = (int32_t)Wdv; g.p_bdv = (int32_t)bdv;
    g.p_Wcv = (int32_t)Wcv; g.p_bcv = (int32_t)bcv; g.p_Wcdv = (int32_t)Wcdv; g.p_bcdv = (int32_t)bcdv;
    g.p_W2_0 = (int32_t)W2(0); g.p_b2_0 = (int32_t)B2(0);
    g.p_W2_stride = NR > 1 ? (int32_t)(W2(1) - W2(0)) : 0;
    g.p_b2_stride = NR > 1 ? (int32_t)(B2(1) - B2(0)) : 0;
    if (p.ws_floats >= (1ll << 31)) { err = "workspace too large for 32-bit offsets"; return TA3N_ERR_INVALID; }

    const int ldZ = NT * NB, ldR = NR * NB, ldF = T * F;
    auto tau = [&](int t, int pos) { return p.tuples[(size_t)t * T + pos]; };

    // ---- GEMM specs (one per affine contraction of the step) ----
    int split_shared_fc = 0;    // (set for the fused step's first launch: ta3n_config.split_k bit 2)
    auto spec_F1 = [&]() {   // shared frame FC + ReLU + dropout_i (models.py:565-575)
        GemmSpec s;
        s.M = BT; s.N = F;
        if (split_shared_fc && D >= 512 && (D / 2) % 128 == 0) {
            // split_k bit 2 (round 6 experiment): the launch has ONE tile per compute unit at the headline shape - one resident workgroup
            // pulls its stages at ~15 B/clk where two pull ~21 - so every tile becomes two workgroups over the two halves of K
            // (EPI_SPLITK: partial tile through L2 + ticket, the second to arrive adds and runs the epilogue)
            s.segs.push_back(mkseg(KC(BASE_X, 0, D), KC(BASE_P, Wsh, D), D / 2));
            s.segs.push_back(mkseg(KC(BASE_X, D / 2, D), KC(BASE_P, Wsh + D / 2, D), D / 2));
            s.split = 2;
        } else
        s.segs.push_back(mkseg(KC(BASE_X, 0, D), KC(BASE_P, Wsh, D), D));
        if (bn_shared) {   // the linear output only: BatchNorm, ReLU and dropout follow in the PH_BN_FWD launch (models.py:565-575)
            s.proto = proto(BASE_WS, g.o_Z0, F);
            with_bias(s.proto, bsh);
            return s;
        }
        s.proto = proto(BASE_WS, g.o_F1, F);
        with_bias(s.proto, bsh);
        s.proto.epi |= EPI_RELU | EPI_DROP_I;
        s.proto.gamma_kind = SK_INV_KEEP_I;
        s.proto.drop_ld = F;
        return s;
    };
    auto spec_Hf = [&]() {   // frame-discriminator hidden layer (models.py:458-459)
        GemmSpec hf;
        hf.M = BT; hf.N = F;
        hf.segs.push_back(mkseg(KC(BASE_WS, g.o_F1, F), KC(BASE_P, Wfd, F), F));
        hf.proto = proto(BASE_WS, g.o_Hf, F);
        with_bias(hf.proto, bfd); hf.proto.epi |= EPI_RELU;
        return hf;
    };
    auto spec_Z = [&](int t) {   // TRN tuple GEMM (TRNmodule.py:60-79): gather+concat folded into the A-operand addressing
        const int j = p.scale_id[t], sl = p.scale_len[t];
        GemmSpec z;
        z.M = B; z.N = NB;
        z.affinity = 300 + j;               // the tuples of scale j read the same W_j
        for (int pos = 0; pos < sl; ++pos)
            z.segs.push_back(mkseg(KC(BASE_WS, g.o_F1 + (int64_t)tau(t, pos) * F, ldF),
                                   KC(BASE_P, trnW(j) + (int64_t)pos * F, sl * F), F));
        z.proto = proto(BASE_WS, g.o_Zr + (int64_t)t * NB, ldZ);
        with_bias(z.proto, trnB(j)); z.proto.epi |= EPI_RELU;
        return z;
    };
    auto spec_Hr = [&](int j) {   // relation-discriminator hidden layer on R_j = sum_t Z_t (models.py:475-479)
        GemmSpec h;
        h.M = B; h.N = NB;
        for (int t = p.tuple_first[j]; t < p.tuple_first[j + 1]; ++t)
            h.segs.push_back(mkseg(KC(BASE_WS, g.o_Zr + (int64_t)t * NB, ldZ), KC(BASE_P, W1(j), NB), NB));
        h.proto = proto(BASE_WS, g.o_Hr + (int64_t)j * NB, ldR);
        with_bias(h.proto, B1(j)); h.proto.epi |= EPI_RELU;
        return h;
    };
    auto spec_Pf = [&]() {   // frame domain logits (models.py:460)
        GemmSpec pf;
        pf.M = BT; pf.N = 2;
        pf.segs.push_back(mkseg(KC(BASE_WS, g.o_Hf, F), KC(BASE_P, Wcd, F), F));
        pf.proto = proto(BASE_WS, g.o_Pf, 2);
        with_bias(pf.proto, bcd);
        return pf;
    };
    auto spec_Y = [&]() {   // video classifier (models.py:686)
        GemmSpec s;
        s.M = B; s.N = C;
        s.segs.push_back(mkseg(KC(BASE_WS, g.o_Vd, NB), KC(BASE_P, Wcv, NB), NB));
        s.proto = proto(BASE_WS, g.o_Y, C);
        with_bias(s.proto, bcv);
        return s;
    };
    auto spec_Y2 = [&]() {   // ens_DA MCD: the second video classifier on the same feature (models.py:717-718)
        GemmSpec s;
        s.M = B; s.N = C;
        s.segs.push_back(mkseg(KC(BASE_WS, g.o_Vd, NB), KC(BASE_P, Wcv2, NB), NB));
        s.proto = proto(BASE_WS, g.o_Y2, C);
        with_bias(s.proto, bcv2);
        return s;
    };
    auto spec_Hv = [&]() {   // video-discriminator hidden layer (models.py:466-467)
        GemmSpec s;
        s.M = B; s.N = NB;
        s.segs.push_back(mkseg(KC(BASE_WS, g.o_Vd, NB), KC(BASE_P, Wdv, NB), NB));
        s.proto = proto(BASE_WS, g.o_Hv, NB);
        with_bias(s.proto, bdv); s.proto.epi |= EPI_RELU;
        return s;
    };
    auto spec_Pv = [&]() {   // video domain logits (models.py:468)
        GemmSpec s;
        s.M = B; s.N = 2;
        s.segs.push_back(mkseg(KC(BASE_WS, g.o_Hv, NB), KC(BASE_P, Wcdv, NB), NB));
        s.proto = proto(BASE_WS, g.o_Pv, 2);
        with_bias(s.proto, bcdv);
        return s;
    };
    auto spec_gHv = [&]() {   // gHv = (gPv Wcdv) * [Hv>0]
        GemmSpec ghv;
        ghv.M = B; ghv.N = NB;
        ghv.segs.push_back(mkseg(KC(BASE_WS, g.o_gPv, 2), KM(BASE_P, Wcdv, NB), 2));
        ghv.proto = proto(BASE_WS, g.o_gHv, NB);
        with_mask(ghv.proto, g.o_Hv, NB);
        return ghv;
    };
    auto spec_gHf = [&]() {   // gHf = (gPf Wcd) * [Hf>0]
        GemmSpec ghf;
        ghf.M = BT; ghf.N = F;
        ghf.segs.push_back(mkseg(KC(BASE_WS, g.o_gPf, 2), KM(BASE_P, Wcd, F), 2));
        ghf.proto = proto(BASE_WS, g.o_gHf, F);
        with_mask(ghf.proto, g.o_Hf, F);
        return ghf;
    };
    // dW = G^T X; bias_dst >= 0: the tiles of the first column block also write db = column sums of G, i.e. the row sums of
    // their own A operand (EPI_ROWSUM_A) - no separate ones^T G tasks, which were as long as a weight-gradient tile each
    auto wgrad = [&](int M, int N, int K, int64_t g_off, int g_ld, int64_t x_off, int x_ld, int64_t dst, int64_t bias_dst = -1) {
        GemmSpec gw;
        gw.M = M; gw.N = N;
        gw.segs.push_back(mkseg(KM(BASE_WS, g_off, g_ld), KM(BASE_WS, x_off, x_ld), K));
        gw.proto = proto(BASE_G, dst, N);
        if (bias_dst >= 0) { gw.proto.epi |= EPI_ROWSUM_A; gw.proto.bias_base = BASE_G; gw.proto.bias_off = (int32_t)bias_dst; }
        return gw;
    };
    auto spec_gVt = [&]() {   // gVt = drop_v'( -beta1 * gHv Wdv + gY Wcv )
        GemmSpec gv;
        gv.M = B; gv.N = NB;
        gv.segs.push_back(mkseg(KC(BASE_WS, g.o_gHv, NB), KM(BASE_P, Wdv, NB), NB, SK_NEG_BETA_VID));
        gv.segs.push_back(mkseg(KC(BASE_WS, g.o_gY, C), KM(BASE_P, Wcv, NB), C));
        if (mcd) gv.segs.push_back(mkseg(KC(BASE_WS, g.o_gY2, C), KM(BASE_P, Wcv2, NB), C));
        gv.proto = proto(BASE_WS, g.o_gVt, NB);
        gv.proto.alpha_kind = SK_REVERSE_MU;      // forward(..., reverse=True): everything behind the video feature sees GradReverse(mu)
        gv.proto.epi |= EPI_DROP_V; gv.proto.gamma_kind = SK_INV_KEEP_V; gv.proto.drop_ld = NB;
        return gv;
    };
    auto spec_gR = [&](int j) {   // gR_j = gRa_j - beta0 * gHr_j W1_j, fanned out through the TRN ReLU masks
        GemmSpec gr;
        gr.M = B; gr.N = NB;
        gr.segs.push_back(mkseg(KC(BASE_WS, g.o_gHr + (int64_t)j * NB, ldR), KM(BASE_P, W1(j), NB), NB));
        gr.proto = proto(BASE_WS, g.o_gR + (int64_t)j * NB, ldR);
        gr.proto.alpha_kind = SK_NEG_BETA_REL;
        with_add(gr.proto, g.o_gRa + (int64_t)j * NB, ldR);
        const int nt = p.tuple_first[j + 1] - p.tuple_first[j];
        gr.proto.fan_count = nt; gr.proto.fan_ld = ldZ;
        for (int k = 0; k < nt; ++k) {
            const int t = p.tuple_first[j] + k;
            gr.proto.fan_mask_off[k] = g.o_Zr + t * NB;
            gr.proto.fan_out_off[k] = g.o_gZ + t * NB;
        }
        return gr;
    };
    auto push_video_head_wgrads = [&](std::vector<GemmSpec> &s) {   // dWcdv, dbcdv, dWcv, dbcv
        s.push_back(wgrad(2, NB, B, g.o_gPv, 2, g.o_Hv, NB, Wcdv, bcdv));
        s.push_back(wgrad(C, NB, B, g.o_gY, C, g.o_Vd, NB, Wcv, bcv));
        if (mcd) s.push_back(wgrad(C, NB, B, g.o_gY2, C, g.o_Vd, NB, Wcv2, bcv2));
    };
    auto push_video_disc_wgrads = [&](std::vector<GemmSpec> &s) {   // dWdv, dbdv
        s.push_back(wgrad(NB, NB, B, g.o_gHv, NB, g.o_Vd, NB, Wdv, bdv));
    };
    auto push_frame_disc_wgrads = [&](std::vector<GemmSpec> &s) {   // dWfd, dbfd
        s.push_back(wgrad(F, F, BT, g.o_gHf, F, g.o_F1, F, Wfd, bfd));
    };
    auto push_relation_level = [&](std::vector<GemmSpec> &s) {   // gR_j (+ fan-out to gZ), dW1_j, db1_j, dW2_j, db2_j
        for (int j = 0; j < NR; ++j) {
            s.push_back(spec_gR(j));
            s.push_back(wgrad(NB, NB, B, g.o_gHr + (int64_t)j * NB, ldR, g.o_R + (int64_t)j * NB, ldR, W1(j), B1(j)));
            s.push_back(wgrad(2, NB, B, g.o_gPrT + (int64_t)j * 2, NR * 2, g.o_Hr + (int64_t)j * NB, ldR, W2(j), B2(j)));
        }
    };
    auto push_trn_wgrads = [&](std::vector<GemmSpec> &s, unsigned scale_mask = ~0u) {   // TRN weight (and bias) gradients of the scales in the mask
        for (int j = 0; j < NR; ++j) {
            if (!((scale_mask >> (j & 31)) & 1)) continue;
            const int sl = T - j;
            for (int pos = 0; pos < sl; ++pos) {   // dW_j[:, pos*F:(pos+1)*F] = sum_t gZ_t^T F1[:, tau_t[pos]]
                GemmSpec gw;
                gw.M = NB; gw.N = F;
                gw.affinity = 100 + j;      // every position of scale j reads the same gZ_t
                for (int t = p.tuple_first[j]; t < p.tuple_first[j + 1]; ++t)
                    gw.segs.push_back(mkseg(KM(BASE_WS, g.o_gZ + (int64_t)t * NB, ldZ),
                                            KM(BASE_WS, g.o_F1 + (int64_t)tau(t, pos) * F, ldF), B));
                gw.proto = proto(BASE_G, trnW(j) + (int64_t)pos * F, sl * F);
                if (pos == 0) { gw.proto.epi |= EPI_ROWSUM_A; gw.proto.bias_base = BASE_G; gw.proto.bias_off = (int32_t)trnB(j); }   // db_j = sum_t column sums of gZ_t
                s.push_back(gw);
            }
        }
    };
    int split_f1_grad = 0;      // (set for the fused step's launches: ta3n_config.split_k)
    auto push_f1_grad = [&](std::vector<GemmSpec> &s) {   // gradient at the frame features (TRN input gradient + frame discriminator's, reversed)
        for (int f = 0; f < T; ++f) {   // gZ1[:, f] = ( -beta2 gHf[:, f] Wfd + sum_{(t,pos): tau_t[pos]==f} gZ_t W_j[:, pos] ) * [F1>0] / keep
            GemmSpec gz;
            gz.M = B; gz.N = F;
            gz.affinity = 200 + f;          // the row panels of frame f read the same weight slabs (Wfd, W_j[:, pos] of every tuple that holds f)
            gz.segs.push_back(mkseg(KC(BASE_WS, g.o_gHf + (int64_t)f * F, ldF), KM(BASE_P, Wfd, F), F, SK_NEG_BETA_FRM));
            for (int t = 0; t < NT; ++t) {
                const int j = p.scale_id[t], sl = p.scale_len[t];
                for (int pos = 0; pos < sl; ++pos)
                    if (tau(t, pos) == f)
                        gz.segs.push_back(mkseg(KC(BASE_WS, g.o_gZ + (int64_t)t * NB, ldZ),
                                                KM(BASE_P, trnW(j) + (int64_t)pos * F, sl * F), NB));
            }
            gz.proto = proto(BASE_WS, g.o_gZ1 + (int64_t)f * F, ldF);
            with_mask(gz.proto, g.o_F1 + (int64_t)f * F, ldF);
            gz.proto.gamma_kind = SK_INV_KEEP_I;
            gz.split = split_f1_grad;
            s.push_back(gz);
        }
    };
    auto push_trn_level = [&](std::vector<GemmSpec> &s) { push_trn_wgrads(s); push_f1_grad(s); };
    auto push_shared_fc_wgrad = [&](std::vector<GemmSpec> &s) {   // shared frame FC weight grad (no input gradient: the features are data)
        GemmSpec gw;
        gw.M = F; gw.N = D;
        gw.segs.push_back(mkseg(KM(BASE_WS, bn_shared ? g.o_gZ0 : g.o_gZ1, F), KM(BASE_X, 0, D), BT));   // (behind the BatchNorm with use_bn)
        gw.proto = proto(BASE_G, Wsh, D);
        gw.proto.epi |= EPI_ROWSUM_A; gw.proto.bias_base = BASE_G; gw.proto.bias_off = (int32_t)bsh;   // dbsh
        s.push_back(gw);
    };

    // ================= forward (group 0) =================
    { std::vector<GemmSpec> s{spec_F1()}; b.add_gemm_phase(0, s); }
    if (bn_shared) b.add_simple_phase(PH_BN_FWD, 0);
    {   // F2
        std::vector<GemmSpec> s{spec_Hf()};
        for (int t = 0; t < NT; ++t) s.push_back(spec_Z(t));
        b.add_gemm_phase(0, s);
    }
    {   // F3
        std::vector<GemmSpec> s;
        for (int j = 0; j < NR; ++j) s.push_back(spec_Hr(j));
        s.push_back(spec_Pf());
        b.add_gemm_phase(0, s);
    }
    b.add_simple_phase(PH_POOL_FWD, 0);   // Pr, attention weights, R, V, Vd
    { std::vector<GemmSpec> s{spec_Y(), spec_Hv()}; if (mcd) s.push_back(spec_Y2()); b.add_gemm_phase(0, s); }   // F6
    { std::vector<GemmSpec> s{spec_Pv()}; b.add_gemm_phase(0, s); }             // F7
    // ================= loss (group 1) =================
    b.add_simple_phase(PH_LOSS, 1);
    // ================= backward (group 2) =================
    {   // Q1: heads that depend only on the logit gradients
        std::vector<GemmSpec> s{spec_gHv(), spec_gHf()};
        push_video_head_wgrads(s);
        s.push_back(wgrad(2, F, BT, g.o_gPf, 2, g.o_Hf, F, Wcd, bcd));   // dWcd = gPf^T Hf, dbcd
        b.add_gemm_phase(2, s);
    }
    {   // Q2: gradient at the pooled video feature + first-layer weight grads of the video/frame discriminators
        std::vector<GemmSpec> s{spec_gVt()};
        push_video_disc_wgrads(s);
        push_frame_disc_wgrads(s);
        b.add_gemm_phase(2, s);
    }
    b.add_simple_phase(PH_POOL_BWD, 2);   // gPrT (attention path), gRa = (1+w) gVt, gHr
    { std::vector<GemmSpec> s; push_relation_level(s); b.add_gemm_phase(2, s); }    // Q5
    { std::vector<GemmSpec> s; push_trn_level(s); b.add_gemm_phase(2, s); }         // Q6
    if (bn_shared) b.add_simple_phase(PH_BN_BWD, 2);
    { std::vector<GemmSpec> s; push_shared_fc_wgrad(s); b.add_gemm_phase(2, s); }   // Q7
    // ================= optimiser (group 3) =================
    b.add_simple_phase(PH_GRAD_NORM, 3);
    b.add_simple_phase(PH_SGD, 3);
    // ================= fused forward + loss + backward (group 4, ta3n_train_step) =================
    // Same arithmetic in 7 launches instead of 15: everything between (Hr, Hf) and (gHr, gHf) - both
    // discriminator heads, the attention pooling, the classifier, the losses and their backward - is one
    // kernel (ta3n_heads.hip); its small weight gradients ride along with the relation level.
    // (The fused heads kernel knows neither the second classifier nor an outside gradient.  use_bn - round 6 - IS part of the fused
    // step: the two BatchNorm launches sit where the unfused lists have them, behind the shared-FC product and in front of its weight
    // gradient: 10 launches instead of 17, models.py:490-543, 569-570; chained launches stay without it.)
    if (heads_supported(NB, C, F) && !mcd && !feat_grads && !(bn_shared && c.chain != 0)) {
        const bool chain = c.chain != 0;
        std::string cerr;
        // chain: the three forward GEMM levels are ONE launch (tile-level hand-offs inside it, Builder::end_chain), likewise the
        // last two backward levels: 5 launches per step instead of 8 (north_star's "gather+concat and the
        // relation MLP as one grouped GEMM launch")
        auto forward_levels = [&](int group, const std::vector<Task> *side) {
            if (chain) b.begin_chain();
            { std::vector<GemmSpec> s{spec_F1()}; b.add_gemm_phase(group, s); }
            if (side) {
                if (chain) b.chain_append(*side);
                else { for (auto &t : *side) p.tasks.push_back(t); p.phases.back().task_count += (int32_t)side->size(); }
            }
            if (!chain && group == 5) return;      // unchained: the pipelined variant only mirrors the first launch
            if (bn_shared) b.add_simple_phase(PH_BN_FWD, group);      // F1 = dropout_i(relu(BatchNorm_domain(Z0))), batch statistics
            {
                std::vector<GemmSpec> s{spec_Hf()};
                for (int t = 0; t < NT; ++t) s.push_back(spec_Z(t));
                b.add_gemm_phase(group, s);
            }
            {
                std::vector<GemmSpec> s;
                for (int j = 0; j < NR; ++j) s.push_back(spec_Hr(j));
                b.add_gemm_phase(group, s);
            }
            if (chain) { const std::string e = b.end_chain(); if (!e.empty()) cerr = e; }
        };
        split_shared_fc = ((c.split_k & 4) && !chain && !bn_shared) ? 1 : 0;
        forward_levels(4, nullptr);
        b.add_simple_phase(PH_HEADS, 4);
        b.sum8[0] = g.o_losses; b.sum8[1] = g.o_loss_part; b.sum8[2] = g.n_vid_wg + g.n_frm_wg;   // logging scalars
        {
            std::vector<GemmSpec> s;
            push_relation_level(s);
            push_video_head_wgrads(s);
            push_video_disc_wgrads(s);
            // dWcd, dbcd = sums over the frame workgroups of their partial sums: exact fp32 column sums in a fixed order
            b.colsum_pending(g.o_fh_part, g.n_frm_wg, 2 * F, 2 * F, Wcd);
            b.colsum_pending(g.o_fh_bpart, g.n_frm_wg, 2, 2, bcd);
            // dWfd: gHf is ready after the heads kernel, so it can fill the CUs this short level leaves idle - unless the
            // next launch reads bf16 twins and this one cannot (odd-shaped head gradients): then it is cheaper there
            const bool twins = (c.flags & (TA3N_FLAG_BF16_MFMA | TA3N_FLAG_F32_SPLIT)) && (c.flags & TA3N_FLAG_BF16_STORE);
            if (!twins) push_frame_disc_wgrads(s);
            b.add_gemm_phase(4, s);
        }
        // The critical path of the backward pass is  relation level -> gradient at F1 -> shared-FC weight gradient; the TRN
        // weight gradients and dWfd feed nothing but the optimiser, so they may ride with either of the last two launches
        // (ta3n_config.wgrads_late, default 0 = with the gradient at F1: measured faster at the headline shape, ta3n_hip.h).
        const bool twins_on = (c.flags & (TA3N_FLAG_BF16_MFMA | TA3N_FLAG_F32_SPLIT)) && (c.flags & TA3N_FLAG_BF16_STORE);
        // wgrads_late: 0 = all with the gradient at F1, 1 = all with the shared-FC weight gradient, otherwise a mask:
        // bit 1 = the frame discriminator's first layer, bit 2 + j = TRN scale j (j = 0: all T frames)
        const unsigned late_mask = c.wgrads_late == 0 ? 0u : c.wgrads_late == 1 ? ~0u : (unsigned)c.wgrads_late;
        const bool late_fd = (late_mask >> 1) & 1;
        const unsigned late_trn = late_mask >> 2;
        if (chain) b.begin_chain();
        split_f1_grad = ((c.split_k & 2) && !chain) ? 2 : 0;
        {
            std::vector<GemmSpec> s;
            if (chain) {      // one launch: the tiles everything else waits for (the gradient at F1) are dispatched first
                push_f1_grad(s);
                if (twins_on && !late_fd) push_frame_disc_wgrads(s);
                push_trn_wgrads(s, ~late_trn);
            } else {
                if (twins_on && !late_fd) push_frame_disc_wgrads(s);
                push_trn_wgrads(s, ~late_trn);
                push_f1_grad(s);
            }
            b.add_gemm_phase(4, s);
        }
        if (bn_shared) b.add_simple_phase(PH_BN_BWD, 4);      // gZ0 + the BatchNorm weight / bias gradients from gZ1
        {
            std::vector<GemmSpec> s;
            push_shared_fc_wgrad(s);
            push_trn_wgrads(s, late_trn);
            if (twins_on && late_fd) push_frame_disc_wgrads(s);
            b.add_gemm_phase(4, s);
        }
        if (chain) { const std::string e = b.end_chain(); if (!e.empty()) cerr = e; }
        {   // group 5: the step's first launch once more, carrying the PREVIOUS step's optimiser update of every parameter
            // it does not read itself (all but the shared frame FC) as EPI_SGD side tasks (ta3n_train_step_after_update)
            const Phase *f1 = nullptr;
            for (const Phase &ph : p.phases)
                if (ph.group == 4 && ph.kind == PH_GEMM) { f1 = &ph; break; }
            b.force_next = f1->wm * 100 + f1->wn * 10 + f1->wk + 1000 * ((f1->bf16 & 15) + ((f1->bf16 & 64) ? 3 : 0)) +
                           10000 * blk_code(f1->rm, f1->rn);
            const int64_t i0 = p.first_floats / 4, i1 = p.live_floats / 4;
            const int n_side = 256;
            const int64_t per = (i1 - i0 + n_side - 1) / n_side;
            std::vector<Task> side;
            for (int k = 0; k < n_side; ++k) {
                Task t;
                std::memset(&t, 0, sizeof(t));
                t.epi = EPI_SGD;
                t.sig = -1;
                t.c_base = BASE_NONE; t.bias_base = BASE_NONE; t.aux_base = BASE_NONE; t.add_base = BASE_NONE;
                t.pad[0] = (int32_t)std::min(i0 + per * k, i1);
                t.pad[1] = (int32_t)std::min(i0 + per * (k + 1), i1);
                side.push_back(t);
            }
            forward_levels(5, &side);
        }
        if (!cerr.empty()) { err = cerr; return TA3N_ERR_INVALID; }
        // every gradient tile of the fused step leaves the sum of its squares in its own slot: the optimiser
        // (ta3n_sgd_step_fused) adds the slots in a fixed order instead of re-reading the gradient buffer
        std::vector<size_t> grad_tasks;
        for (const Phase &ph : p.phases)
            if (ph.group == 4 && ph.kind == PH_GEMM)
                for (int i = ph.task_begin; i < ph.task_begin + ph.task_count; ++i)
                    if ((p.tasks[i].seg_count > 0 || (p.tasks[i].epi & EPI_COLSUM)) && p.tasks[i].c_base == BASE_G) grad_tasks.push_back((size_t)i);
        // use_bn: the BatchNorm weight / bias gradients come from the PH_BN_BWD launch, not from a tile - its workgroups ((F + BN_COLS - 1) / BN_COLS
        // column blocks x 2 domains) leave their sums of squares in the LAST slots of the same region (bn_shared_bwd_kernel)
        const int n_bn_slots = bn_shared ? 2 * ((F + BN_COLS - 1) / BN_COLS) : 0;
        g.n_sumsq = (int32_t)grad_tasks.size() + n_bn_slots;
        g.o_sumsq = (int32_t)b.add_region("sumsq", g.n_sumsq);
        for (size_t k = 0; k < grad_tasks.size(); ++k) {
            p.tasks[grad_tasks[k]].epi |= EPI_SUMSQ;
            p.tasks[grad_tasks[k]].pad[3] = g.o_sumsq + (int32_t)k;
        }
        // (the heads kernel keeps the twin of gHf; gZ and gZ1 are read by GEMM launches only: twin-only when those read twins.
        // use_bn: the BatchNorm launches keep the twins of what they produce - F1 and gZ0 - and READ gZ1 in fp32)
        std::vector<Span> kept{Span{g.o_gHf, g.o_gHf + (int64_t)BT * F}};
        std::vector<Span> gemm_only{Span{g.o_gZ, g.o_gZ + (int64_t)B * NT * NB}};
        if (bn_shared) {
            kept.push_back(Span{g.o_F1, g.o_F1 + (int64_t)BT * F});
            kept.push_back(Span{g.o_gZ0, g.o_gZ0 + (int64_t)BT * F});
        } else {
            gemm_only.push_back(Span{g.o_gZ1, g.o_gZ1 + (int64_t)BT * F});
        }
        add_bf16_twins(p, b, g, BT, D, kept, gemm_only);
        if (p.ws_floats >= (1ll << 31)) { err = "workspace too large for 32-bit offsets"; return TA3N_ERR_INVALID; }
    } else {
        // no fused step (TA3N_FLAG_FEATURE_GRADS: dis_DA DAN / JAN put a gradient between forward and backward; TA3N_FLAG_MCD; a shape the heads
        // kernel does not cover): the unfused lists are all there is - round 6: they read twins too (add_bf16_twins: the unfused family)
        add_bf16_twins(p, b, g, BT, D, {}, {});
        if (p.ws_floats >= (1ll << 31)) { err = "workspace too large for 32-bit offsets"; return TA3N_ERR_INVALID; }
    }
    if (b.mixed_kinds) { err = "internal: a GEMM spec mixes operand kinds across its K segments"; return TA3N_ERR_INVALID; }
    if (b.n_split_pairs > 0) {      // split-K pairs: partial tiles and tickets behind everything else (outside the span the twins mirror)
        std::vector<int64_t> pair_off(b.n_split_pairs, -1);
        int64_t floats = 0;
        for (const Phase &ph : p.phases) {
            if (ph.kind != PH_GEMM) continue;
            const int64_t tile = (int64_t)(32 * ph.wm * std::max(ph.rm, 1)) * (32 * ph.wn * std::max(ph.rn, 1));
            for (int i = ph.task_begin; i < ph.task_begin + ph.task_count; ++i) {
                const Task &t = p.tasks[i];
                if ((t.epi & EPI_SPLITK) && pair_off[t.pad[0]] < 0) { pair_off[t.pad[0]] = floats; floats += 2 * tile; }
            }
        }
        const int64_t o_part = b.add_region("splitk_part", floats);
        const int64_t o_tick = b.add_region("splitk_ticket", b.n_split_pairs);
        for (auto &t : p.tasks)
            if (t.epi & EPI_SPLITK) {
                const int pair = t.pad[0];
                t.pad[0] = (int32_t)(o_part + pair_off[pair]);
                t.pad[1] = (int32_t)(o_tick + pair);
            }
        if (p.ws_floats >= (1ll << 31)) { err = "workspace too large for 32-bit offsets"; return TA3N_ERR_INVALID; }
    }
    for (auto &t : p.tasks)      // (after the twin re-addressing: the copies must be the final Segs)
        if (t.seg_count > 0) t.seg0 = p.segs[t.seg_begin];
    return TA3N_OK;
}

// Register-blocked tiles exist for launches that read bf16 twins only, and whether a launch does is known once the whole
// plan is laid out (add_bf16_twins): build, look, and rebuild without the blocking where it cannot be used.
int ta3n::build_plan(ta3n_plan &p, std::string &err) {
    const ta3n_config cfg = p.cfg;
    uint64_t deny = 0;
    for (int attempt = 0; attempt < 16; ++attempt) {
        p = ta3n_plan();
        p.cfg = cfg;
        p.deny_blocking = deny;
        const int rc = build_plan_once(p, err);
        if (rc != TA3N_OK) return rc;
        uint64_t bad = 0;
        for (size_t i = 0; i < p.phases.size(); ++i) {
            const Phase &ph = p.phases[i];
            if (ph.kind == PH_GEMM && (ph.rm * ph.rn > 1 || (ph.bf16 & 64)) && !(ph.bf16 & 16)) bad |= 1ull << (i & 63);
        }
        if (bad == 0) return TA3N_OK;
        deny |= bad;
    }
    err = "internal: register-blocked tile selection did not settle";
    return TA3N_ERR_INVALID;
}
