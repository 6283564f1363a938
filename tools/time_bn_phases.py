"""Per-launch times (HIP events, each launch alone) of the fused step with use_bn AdaBN at the headline shape: where the BatchNorm variant's extra time goes.
usage (GPU box): python tools/time_bn_phases.py [bf16|f32]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ta3n_amd.engine import TrainEngine
from ta3n_amd.synthetic import synth_batch, synth_state
bf16 = (sys.argv[1] if len(sys.argv) > 1 else "bf16") == "bf16"
for bn in ("none", "AdaBN"):
    eng = TrainEngine(128, 74, 5, 2048, 512, 12, bf16=bf16, bf16_store=bf16, use_bn=bn)
    eng.load_state(synth_state({n: s for n, _, s, _ in eng.plan.params}, seed=7, scale="init"))
    xs, xt, ys, yt = synth_batch(12, 5, 2048, 128, 74, seed=1234)
    eng.set_batch(xs.cuda(), xt.cuda(), ys.cuda())
    eng.set_hyper([0.75, 0.75, 0.5], 0.003, 0.03)
    for _ in range(3):
        eng.train_step([0.75, 0.75, 0.5], 0.003, 0.03)
    torch.cuda.synchronize()
    ph = eng.time_phases(20)
    names = {0: "gemm", 5: "sgd", 6: "heads", 10: "bn_fwd", 11: "bn_bwd", 4: "norm"}
    print(f"use_bn={bn} {'bf16' if bf16 else 'f32'}:", " ".join(f"{names.get(k, k)}[{t}]={1e3 * ms:.1f}" for k, t, n, ms in ph), " sum", round(1e3 * sum(p[3] for p in ph), 1))
