"""Tile codes of the UNFUSED launch lists now that five of their launches read twins (round 6): candidates for tuning.py entries 0-9 at the headline shape.
usage (GPU box): python tools/tune_unfused_twins.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ta3n_amd.engine import TrainEngine
from ta3n_amd.synthetic import synth_batch, synth_state
from ta3n_amd.tuning import TUNED
base = list(TUNED[(202, 5, 2048, 512, "bf16")])
xs, xt, ys, yt = synth_batch(12, 5, 2048, 128, 74, seed=1234)
cands = {"shipped": base}
for name, edits in (("Q6=2222", {8: 2222}), ("Q7=2222", {9: 2222}), ("Q6,Q7=2222", {8: 2222, 9: 2222}), ("F3=3124", {2: 3124}), ("F3=3124,Q6,Q7=2222", {2: 3124, 8: 2222, 9: 2222}),
                    ("Q6=12222", {8: 12222}), ("F2=2222", {1: 2222}), ("Q6,Q7=2222,F2=2214", {1: 2214, 8: 2222, 9: 2222})):
    t = list(base)
    for k, v in edits.items():
        t[k] = v
    cands[name] = t
for kind, kw in (("unfused", dict(fused=False)), ("DAN", dict(dis_DA="DAN", alpha=0.5))):
    for name, tiles in cands.items():
        try:
            eng = TrainEngine(128, 74, 5, 2048, 512, 12, dropout_i=0.5, dropout_v=0.5, clip=20.0, bf16=True, bf16_store=True, phase_tiles=tiles, **kw)
        except Exception as ex:      # noqa: BLE001
            print(kind, name, "->", type(ex).__name__, str(ex)[:120]); continue
        eng.load_state(synth_state({n: s for n, _, s, _ in eng.plan.params}, seed=7, scale="trained"))
        eng.set_batch(xs.cuda(), xt.cuda(), ys.cuda())
        for _ in range(30):
            eng.train_step([0.75, 0.75, 0.5], 0.003, 1e-3)
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(100):
                eng.train_step([0.75, 0.75, 0.5], 0.003, 1e-3)
            torch.cuda.synchronize()
            best = min(best, 1e6 * (time.perf_counter() - t0) / 100)
        ph = [round(1e3 * ms, 1) for k, t, n, ms in eng.time_phases(10) if k == 0]
        print(f"{kind:8s} {name:22s} {best:6.1f} us/step   GEMM launches {ph}")
