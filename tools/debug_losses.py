import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ta3n_amd.engine import TrainEngine
from ta3n_amd.synthetic import synth_batch, synth_state
for graph in (False, True):
    eng = TrainEngine(128, 74, 5, 2048, 512, 12, dropout_i=0.5, dropout_v=0.5)
    eng.load_state(synth_state({n: s for n, _, s, _ in eng.plan.params}, seed=7, scale="init"))
    xs, xt, ys, yt = synth_batch(12, 5, 2048, 128, 74, seed=1)
    eng.set_batch(xs.cuda(), xt.cuda(), ys.cuda())
    if graph:
        eng.set_hyper([0.75, 0.75, 0.5], 0.003, 3e-2)
        eng.capture()
    for i in range(4):
        eng.train_step([0.75, 0.75, 0.5], 0.003, 3e-2, valid_source=128, valid_target=74, global_source=128, global_target=74)
        torch.cuda.synchronize()
        print("graph" if graph else "eager", i, eng.region("losses").tolist(), eng.region("grad_norm").tolist()[:2], flush=True)
