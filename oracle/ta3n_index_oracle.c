/*
 * CPU oracle for the INTEGER part of the TA3N hot path (bit-exact contract).
 * TEST INFRASTRUCTURE ONLY: built by oracle/Makefile into oracle/_build/,
 * loaded only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg.  The product (ta3n_amd/) never links or loads it.
 *
 * Restates, in the reference's own (enumerate-everything) way:
 *   - TRNmodule.py:34-41, 84-86  all C(T,s) frame tuples per scale in
 *     itertools.combinations (lexicographic) order;
 *   - TRNmodule.py:60, 68-71     scale 0 uses tuple 0, later scales use
 *     idx_i = int(ceil(i * n_total / n_select)), n_select = min(3, n_total);
 *   - dataset.py:103-116         _get_test_indices (1-based frame ids).
 * The product generates the same tuples by combinatorial unranking
 * (ta3n_amd/csrc/ta3n_index.cpp); tests compare the two for T = 2..25.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#define SUBSAMPLE_NUM 3

/* lexicographic successor of a k-combination of {0..n-1}; returns 0 at the end */
static int next_comb(int *c, int n, int k) {
    int i = k - 1;
    while (i >= 0 && c[i] == n - k + i) --i;
    if (i < 0) return 0;
    ++c[i];
    for (int j = i + 1; j < k; ++j) c[j] = c[j - 1] + 1;
    return 1;
}

/* Writes the selected tuples scale by scale (scale T first).  tuples is a
 * [n_out][T] row-major array padded with -1; scale_len[r] is the tuple size of
 * row r; scale_id[r] the scale index (0 = T-frame scale).  Returns n_out. */
int ta3n_oracle_relation_table(int T, int32_t *tuples, int32_t *scale_len, int32_t *scale_id) {
    int n_out = 0;
    int *c = (int *)malloc(sizeof(int) * (size_t)T);
    for (int sid = 0, s = T; s >= 2; --s, ++sid) {
        /* count C(T,s) by enumeration, exactly like len(list(combinations)) */
        long n_total = 0;
        for (int j = 0; j < s; ++j) c[j] = j;
        do { ++n_total; } while (next_comb(c, T, s));
        int n_sel = (sid == 0) ? 1 : (int)(n_total < SUBSAMPLE_NUM ? n_total : SUBSAMPLE_NUM);
        for (int i = 0; i < n_sel; ++i) {
            long idx = (sid == 0) ? 0 : (long)ceil((double)(i * n_total) / (double)n_sel);
            for (int j = 0; j < s; ++j) c[j] = j;
            for (long r = 0; r < idx; ++r) next_comb(c, T, s);
            for (int j = 0; j < T; ++j) tuples[(size_t)n_out * T + j] = (j < s) ? c[j] : -1;
            scale_len[n_out] = s;
            scale_id[n_out] = sid;
            ++n_out;
        }
    }
    free(c);
    return n_out;
}

/* dataset.py:103-116.  Returns 0 on success, -1 where the reference raises
 * (num_select <= 0). */
int ta3n_oracle_test_indices(int num_frames, int num_segments, int new_length, int64_t *out) {
    int num_min = num_segments + new_length - 1;
    int num_select = num_frames - new_length + 1;
    if (num_frames >= num_min) {
        double tick = (double)num_select / (double)num_segments;
        for (int x = 0; x < num_segments; ++x) out[x] = (int64_t)(tick / 2.0 + tick * (double)x) + 1;
        return 0;
    }
    if (num_select <= 0) return -1;
    for (int x = 0; x < num_segments; ++x) out[x] = (x < num_select ? x : num_select - 1) + 1;
    return 0;
}
