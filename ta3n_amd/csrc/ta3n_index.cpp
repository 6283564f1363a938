// Integer layer of the TA3N hot path (host code, bit-exact contract).
//
//  * relation tuples of RelationModuleMultiScale (reference TRNmodule.py:30-41,
//    60, 68-71, 84-86).  The reference enumerates every C(T,s) combination with
//    itertools and then indexes the list; here the idx-th lexicographic
//    combination is produced directly by combinatorial unranking, so building
//    the table is O(T^2) instead of O(C(T,T/2)).
//  * TSNDataSet._get_test_indices (reference dataset.py:103-116).
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../include/ta3n_hip.h"

namespace {

constexpr int kSubsample = 3;  // TRNmodule.py:32

// C(n,k) in unsigned 64-bit; saturates at UINT64_MAX (never reached for T <= 64
// on the paths used here, guarded by the caller).
uint64_t binom(int n, int k) {
    if (k < 0 || k > n) return 0;
    if (k > n - k) k = n - k;
    unsigned __int128 r = 1;
    for (int i = 1; i <= k; ++i) {
        r = r * (unsigned)(n - k + i) / (unsigned)i;
        if (r > (unsigned __int128)UINT64_MAX) return UINT64_MAX;
    }
    return (uint64_t)r;
}

// idx-th (0-based) k-combination of {0..n-1} in lexicographic order.
void unrank(int n, int k, uint64_t idx, int32_t *out) {
    int x = 0;
    for (int i = 0; i < k; ++i) {
        // choose the smallest x such that the number of combinations starting
        // with a smaller element at position i is <= idx
        for (;; ++x) {
            uint64_t cnt = binom(n - x - 1, k - i - 1);
            if (idx < cnt) break;
            idx -= cnt;
        }
        out[i] = x;
        ++x;
    }
}

}  // namespace

extern "C" int ta3n_num_relation_tuples(int T) {
    if (T < 2 || T > 64) return TA3N_ERR_INVALID;
    int n = 1;
    for (int s = T - 1; s >= 2; --s) {
        uint64_t c = binom(T, s);
        n += (int)(c < (uint64_t)kSubsample ? c : (uint64_t)kSubsample);
    }
    return n;
}

extern "C" int ta3n_relation_table(int T, int32_t *tuples, int32_t *scale_len, int32_t *scale_id) {
    if (T < 2 || T > 64 || !tuples || !scale_len || !scale_id) return TA3N_ERR_INVALID;
    int n_out = 0;
    for (int sid = 0, s = T; s >= 2; --s, ++sid) {
        const uint64_t n_total = binom(T, s);
        const int n_sel = (sid == 0) ? 1 : (int)(n_total < (uint64_t)kSubsample ? n_total : (uint64_t)kSubsample);
        for (int i = 0; i < n_sel; ++i) {
            // TRNmodule.py:71  int(ceil(i * num_total / num_select)) in Python float arithmetic
            uint64_t idx = 0;
            if (sid != 0) idx = (uint64_t)std::ceil((double)((uint64_t)i * n_total) / (double)n_sel);
            int32_t *row = tuples + (size_t)n_out * T;
            unrank(T, s, idx, row);
            for (int j = s; j < T; ++j) row[j] = -1;
            scale_len[n_out] = s;
            scale_id[n_out] = sid;
            ++n_out;
        }
    }
    return n_out;
}

extern "C" int ta3n_segment_indices(int num_frames, int num_segments, int new_length, int64_t *out) {
    if (num_segments <= 0 || new_length <= 0 || !out) return TA3N_ERR_INVALID;
    const int num_min = num_segments + new_length - 1;
    const int num_select = num_frames - new_length + 1;
    if (num_frames >= num_min) {
        const double tick = (double)num_select / (double)num_segments;
        for (int x = 0; x < num_segments; ++x) out[x] = (int64_t)(tick / 2.0 + tick * (double)x) + 1;
        return TA3N_OK;
    }
    if (num_select <= 0) return TA3N_ERR_INVALID;  // the reference indexes an empty array here
    for (int x = 0; x < num_segments; ++x) out[x] = (int64_t)(x < num_select ? x : num_select - 1) + 1;
    return TA3N_OK;
}
