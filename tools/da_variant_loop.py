"""One DA variant's train loop for a profiler (headline shape, bf16 twins): python tools/da_variant_loop.py {adabn|mcd|dan|jan} [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ta3n_amd.engine import TrainEngine
from ta3n_amd.synthetic import synth_batch, synth_state
which, steps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 50
kw = dict(adabn=dict(use_bn="AdaBN"), mcd=dict(ens_DA="MCD", mu=0.5), dan=dict(dis_DA="DAN", alpha=0.5), jan=dict(dis_DA="JAN", alpha=0.5, place_dis=("Y", "Y", "N")))[which]
eng = TrainEngine(128, 74, 5, 2048, 512, 12, dropout_i=0.5, dropout_v=0.5, clip=20.0, bf16=True, bf16_store=True, **kw)
eng.load_state(synth_state({n: s for n, _, s, _ in eng.plan.params}, seed=7, scale="trained"))
xs, xt, ys, yt = synth_batch(12, 5, 2048, 128, 74, seed=1234)
eng.set_batch(xs.cuda(), xt.cuda(), ys.cuda())
for _ in range(steps):
    eng.train_step([0.75, 0.75, 0.5], 0.003, 1e-3)
torch.cuda.synchronize()
print(which, "finite", bool(torch.isfinite(eng.P).all()))
