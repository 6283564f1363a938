"""Measured GEMM tile choices per launch for the shapes this repository benchmarks (MI355X, `bench.py --autotune`;
engine.autotune_phase_tiles does the measurement: every candidate tile for every launch, HIP events on the launch stream).

A tile code is WM*100 + WN*10 + WK (waves of the workgroup in M, N and the in-workgroup K split), + 1000 * LDS stages for the
bf16 kernels (5000 / 6000 / 7000: two / three / four HALF stages of 64 k, bf16-twin kernel), + 10000 / 20000 / 30000 for 2 row / 2 column / 2 x 2
32x32 blocks per wave (bf16-twin kernel: 128x64, 64x128, 128x128 tiles; 46221 / 56221: 192x128 / 256x128, four waves, half stages).  Entries 0-9 are the forward / loss / backward launches of the unfused sequence, 10-15 the six GEMM launches of
the fused step (ta3n_train_step); 0 = the plan builder's own choice (ta3n_plan.cpp: add_gemm_phase).

What decides a launch ("tile choice"): a CU fills its LDS at ~41 B/clk whatever the tile, so a launch wants (a) at
least one workgroup per CU and (b) beyond that the largest tile, which brings the fewest operand bytes per flop; launches with
few, long tiles want a third LDS stage, launches with two workgroups per CU do not (they hide each other's latency)."""
from __future__ import annotations

from typing import List, Optional

# (videos per step, segments, feature_dim, fc_dim, arithmetic) -> per-launch tile codes
TUNED = {
    # BASELINE configs[1] / [2]: UCF->HMDB_full, 128 + 74 videos, 5 segments, 2048-d
    # (entries 10-15 by the time of the whole pipelined step, tools/tune_in_sequence.py: consecutive launches of the SAME kernel
    # instantiation are ~1 us cheaper each than a change of kernel - 3124 for the three forward levels beats their individually
    # fastest tiles 3124 / 2214 / 2118 by 2.7 us per step)
    (202, 5, 2048, 512, "bf16"): [3124, 3124, 2118, 2118, 2118, 2118, 2118, 2124, 2122, 2124, 3124, 3124, 3124, 2124, 2222, 2222],
    (202, 5, 2048, 512, "f32"): [124, 118, 118, 118, 118, 118, 118, 124, 124, 222, 124, 114, 118, 124, 124, 124],
    # the same shape, fp32-grade contractions as three bf16 MFMAs on split operands (TA3N_FLAG_F32_SPLIT): fp32 stage images
    (202, 5, 2048, 512, "f32x3"): [3124, 3114, 2118, 2118, 2118, 2118, 2118, 2124, 2122, 2124, 3124, 3114, 2118, 3124, 2212, 2124],
    # ... and with "pair twins" (TA3N_FLAG_F32_SPLIT | _BF16_STORE: the producers store the hi and the lo plane): bf16 stage images of 64 k
    (202, 5, 2048, 512, "f32x3p"): [3124, 3114, 2118, 2118, 2118, 2118, 2118, 2124, 2122, 2124, 3214, 3214, 3214, 2124, 2122, 2122],
    # BASELINE configs[3]: 512 + 512 videos, 9 segments, 2048-d, 30 classes
    # (round 4: entry 10, the shared-FC launch, measured on the half-stage kernel 7222 - 62.6 -> 46.1 us timed alone, 451.6 -> 445.8 / 449.8 ->
    # 432.4 us per forward+backward step on two boxes - and then under bench.py's protocol, the pipelined step whose first launch also
    # carries the update's side workgroups: 0.492-0.504 ms against 0.493-0.496 for 32222, three alternating processes each.  Not adopted:
    # profiles/r04_half_stage_ab.txt.)
    # (round 5: entry 10 on 35221 - the 128x128 tile with FOUR waves on two half stages, two workgroups per compute unit: the launch's 288
    # tiles are resident at once instead of in two rounds of 256 + 32.  Under bench.py's protocol, alternating: launch 70.3 -> 53.0 us, step
    # 0.4658 / 0.4680 -> 0.4556 / 0.4604 ms; the same tile on launches 11 and 14 measured slower (74.6 -> 82.0, 129.5 -> 147.5 us):
    # profiles/r05_heads_tiles_ab.txt)
    (1024, 9, 2048, 512, "bf16"): [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 35221, 32222, 2222, 2222, 32222, 3222],
    # BASELINE configs[4] (128 + 128 videos, 12 segments, 1024-d, two streams): NO entry - the plan's heuristic.  Round 3 shipped a list
    # chosen on a single-stream 200-step sweep (277.2 -> 271.7 us, one run each); under the protocol the configuration is judged by
    # (two concurrent streams, 20 steps after 5, five processes each: profiles/r04_config5_protocol.txt) it measures 0.516-0.523 ms
    # against 0.490-0.499 ms for the heuristic - reverted.
}


def tuned_phase_tiles(batch: int, num_segments: int, feature_dim: int, fc_dim: int, bf16: bool, twins: bool,
                      split: bool = False) -> Optional[List[int]]:
    """The measured list for this shape, or None (the plan builder's heuristic then picks per launch)."""
    arith = ("f32x3p" if twins else "f32x3") if split else ("bf16" if bf16 else "f32")
    key = (int(batch), int(num_segments), int(feature_dim), int(fc_dim), arith)
    t = TUNED.get(key)
    if t is None:
        # nearest measured shape (an exact-key table is brittle - 200 or 208 videos per step fell to the heuristic): same
        # segments, widths and arithmetic - they fix every launch's K loops and tile grid columns - and a batch within 3/4 .. 4/3 of a
        # measured one, i.e. the same number of row tiles per CU to within one; the closest batch wins.
        near = [(abs(k[0] - batch), k) for k in TUNED if k[1:] == key[1:] and 3 * k[0] <= 4 * batch and 3 * batch <= 4 * k[0]]
        if not near:
            return None
        t = TUNED[min(near)[1]]
    if bf16 and not twins:      # register-blocked tiles exist for the twin kernel only (the plan would drop them anyway)
        t = [c % 10000 for c in t]
    return list(t)
