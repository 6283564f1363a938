"""Step time of the DA options that run as unfused launch lists (ens_DA MCD, dis_DA DAN / JAN) beside the fused step, headline shape, bf16 twins and fp32.
usage (GPU box): python tools/time_da_variants.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ta3n_amd.engine import TrainEngine
from ta3n_amd.synthetic import synth_batch, synth_state
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
Bs, Bt, T, D, F, C = 128, 74, 5, 2048, 512, 12
xs, xt, ys, yt = synth_batch(C, T, D, Bs, Bt, seed=1234)
for bf16 in (True, False):
    for name, kw in (("fused step", {}), ("unfused lists", dict(fused=False)), ("ens_DA MCD", dict(ens_DA="MCD", mu=0.5)),
                     ("dis_DA DAN", dict(dis_DA="DAN", alpha=0.5)), ("dis_DA JAN", dict(dis_DA="JAN", alpha=0.5, place_dis=("Y", "Y", "N")))):
        try:
            eng = TrainEngine(Bs, Bt, T, D, F, C, dropout_i=0.5, dropout_v=0.5, clip=20.0, bf16=bf16, bf16_store=bf16, **kw)
        except Exception as ex:      # noqa: BLE001
            print(f"{name}: {type(ex).__name__}: {ex}"[:200]); continue
        eng.load_state(synth_state({n: s for n, _, s, _ in eng.plan.params}, seed=7, scale="init"))
        eng.set_batch(xs.cuda(), xt.cuda(), ys.cuda())
        for _ in range(30):
            eng.train_step([0.75, 0.75, 0.5], 0.003, 0.03)
        best = 1e9
        for rep in range(3):      # best of three runs: the first run of a configuration also pays one-time costs (code objects, allocator)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(steps):
                eng.train_step([0.75, 0.75, 0.5], 0.003, 0.03)
            torch.cuda.synchronize()
            best = min(best, 1e6 * (time.perf_counter() - t0) / steps)
        print(f"{'bf16' if bf16 else 'f32 '} {name:14s} fused={eng.fused}: {best:.0f} us/step (one library call per launch group, host included; best of 3 x {steps})")
