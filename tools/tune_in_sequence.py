"""Tile choice per GEMM launch of the fused step by the time of the WHOLE pipelined step (coordinate descent over the six launches),
not by each launch's time in isolation (bench.py --autotune / engine.autotune_phase_tiles): a launch's tile shape also decides how its
successor finds the caches and the CUs.  Headline shape; prints the table and the final list for ta3n_amd/tuning.py.
usage: python tools/tune_in_sequence.py [bf16|f32|f32x3] [sweeps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ta3n_amd.engine import TrainEngine
from ta3n_amd.synthetic import synth_batch, synth_state
from ta3n_amd.tuning import tuned_phase_tiles

arith = sys.argv[1] if len(sys.argv) > 1 else "bf16"
sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
kw = dict(bf16=True, bf16_store=True) if arith == "bf16" else dict(f32_split=True, bf16_store=True) if arith == "f32x3" else {}
Bs, Bt, T, D, F, C = (int(v) for v in os.environ.get("TA3N_TUNE_SHAPE", "128,74,5,2048,512,12").split(","))
base = tuned_phase_tiles(Bs + Bt, T, D, F, arith == "bf16", arith != "f32", split=(arith == "f32x3")) or [0] * 16
stages = (2, 3) if arith != "f32" else (0,)
CANDS = [s * 1000 + c for c in (114, 118, 212, 122, 214, 124, 221, 222) for s in stages]
if arith == "bf16" and Bs + Bt >= 512:      # register-blocked tiles of the twin kernel pay at the larger shapes
    CANDS = [c for c in CANDS if c % 1000 in (214, 124, 221, 222)] + [12222, 13222, 22222, 23222, 32222, 32221, 35221, 36222, 6222]
if os.environ.get("TA3N_TUNE_CANDS"):         # bounded GPU time: an explicit candidate list / launch order
    CANDS = [int(v) for v in os.environ["TA3N_TUNE_CANDS"].split(",")]
LAUNCHES = [int(v) for v in os.environ.get("TA3N_TUNE_LAUNCHES", "10,11,12,13,14,15").split(",")]
xs, xt, ys, yt = synth_batch(C, T, D, Bs, Bt, seed=1)
xs, xt, ys = xs.cuda(), xt.cuda(), ys.cuda()
sched = [([0.75, 0.75, 0.5], 0.003, 1e-3)] * 200


def step_us(tiles, reps=3, stats=False):
    eng = TrainEngine(Bs, Bt, T, D, F, C, dropout_i=0.5, dropout_v=0.5, phase_tiles=tiles, **kw)
    shapes = {n: s for n, _, s, _ in eng.plan.params}
    eng.load_state(synth_state(shapes, seed=7, scale="init"))
    eng.set_batch(xs, xt, ys)
    eng.train_steps(sched[:40]); torch.cuda.synchronize()
    runs = []
    for _ in range(reps):
        t0 = time.perf_counter()
        eng.train_steps(sched)
        eng.flush(); torch.cuda.synchronize()
        runs.append((time.perf_counter() - t0) / len(sched) * 1e6)
    if stats:                                   # (median, spread = max - min) over the repeats
        runs.sort()
        return runs[len(runs) // 2], runs[-1] - runs[0]
    return min(runs)


if os.environ.get("TA3N_TUNE_COMBOS"):      # "3124,3214,2118;3124,3124,2118;...": the first fused launches' tiles, alternated three times
    combos = [[int(v) for v in c.split(",")] for c in os.environ["TA3N_TUNE_COMBOS"].split(";")]
    for rep in range(3):
        for c in combos:
            t = list(base); t[10:10 + len(c)] = c
            print(rep, c, f"{step_us(t, reps=4):.2f} us", flush=True)
    sys.exit(0)
cur = list(base)
print("base", cur[10:16], f"{step_us(cur):.2f} us", flush=True)
for sw in range(sweeps):
    for ph in LAUNCHES:
        table = {}
        for c in CANDS:
            t = list(cur); t[ph] = c
            try:
                table[c] = step_us(t, reps=2)
            except Exception as ex:      # noqa: BLE001 - a tile the plan rejects for this launch
                table[c] = float("inf")
        best = min(table, key=table.get)
        print(f"sweep {sw} launch {ph}: " + "  ".join(f"{c}:{v:.1f}" for c, v in table.items()) + f"  -> {best} (was {cur[ph]})", flush=True)
        if best == cur[ph]:
            continue
        # Guard (a81f12a shipped a list whose total gain was inside its own noise and lost 4 % under the judged protocol):
        # a challenger replaces the incumbent only if, re-measured with >= 3 repeats each in fresh engines, its MEDIAN wins by more than
        # twice the larger of the two run-to-run spreads.
        t_new = list(cur); t_new[ph] = best
        (m_new, s_new), (m_old, s_old) = step_us(t_new, reps=4, stats=True), step_us(cur, reps=4, stats=True)
        ok = m_new < m_old - 2 * max(s_new, s_old)
        print(f"   challenger {best}: median {m_new:.2f} us (spread {s_new:.2f}) vs incumbent {cur[ph]}: {m_old:.2f} (spread {s_old:.2f}) -> "
              f"{'ACCEPT' if ok else 'keep the incumbent'}", flush=True)
        if ok:
            cur[ph] = best
print("final", cur, "median %.2f us (spread %.2f) vs base %.2f (spread %.2f)" % (*step_us(cur, reps=5, stats=True), *step_us(base, reps=5, stats=True)))
print("NOTE: a list for a multi-stream configuration must be confirmed under that configuration's own protocol (bench.py --config 5 --steps 20 "
      "--warmup 5, >= 5 processes) before it goes into ta3n_amd/tuning.py")
