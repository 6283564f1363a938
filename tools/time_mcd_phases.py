"""Where an ens_DA MCD step spends its time (headline shape): every stage of TrainEngine._enqueue_step timed on its own (host clock, a synchronize after each).
usage (GPU box): python tools/time_mcd_phases.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ta3n_amd.engine import TrainEngine
from ta3n_amd.synthetic import synth_batch, synth_state
xs, xt, ys, yt = synth_batch(12, 5, 2048, 128, 74, seed=1234)
for bf16 in (False, True):
    eng = TrainEngine(128, 74, 5, 2048, 512, 12, dropout_i=0.5, dropout_v=0.5, clip=20.0, ens_DA="MCD", mu=0.5, bf16=bf16, bf16_store=bf16)
    eng.load_state(synth_state({n: s for n, _, s, _ in eng.plan.params}, seed=7, scale="init"))
    eng.set_batch(xs.cuda(), xt.cuda(), ys.cuda())
    for _ in range(5):
        eng.train_step([0.75, 0.75, 0.5], 0.003, 0.03)
    stages = [("set_hyper", lambda: eng.set_hyper([0.75, 0.75, 0.5], 0.003, 0.03, train=True)), ("forward", eng.forward), ("loss", eng.loss),
              ("mcd_source_loss", eng.mcd_source_loss), ("mcd_second_forward", eng.mcd_second_forward), ("backward", eng.backward),
              ("mcd_second_backward", eng.mcd_second_backward), ("sgd_step", eng.sgd_step)]
    acc = {k: 0.0 for k, _ in stages}
    for rep in range(20):
        for k, fn in stages:
            torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); acc[k] += time.perf_counter() - t0
    print("bf16" if bf16 else "f32 ", " ".join(f"{k}={1e6 * v / 20:.0f}" for k, v in acc.items()), " sum", round(1e6 * sum(acc.values()) / 20))
