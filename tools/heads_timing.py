"""Stage timeline of the fused heads kernel (workgroup 0): build the library with -DTA3N_HEADS_TIMING, run one
fused step, print s_memtime deltas between stage boundaries.  Debug aid, not part of the product.
usage: python tools/heads_timing.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ta3n_amd import build
build.build(force=True, verbose=False, extra_flags=("-DTA3N_HEADS_TIMING",))
from ta3n_amd.engine import TrainEngine
eng = TrainEngine(128, 74, 5, 2048, 512, 12)
eng.X.uniform_(0, 1)
for v in eng.param_views().values(): v.normal_(0, 0.02)
eng.set_hyper([0.75, 0.75, 0.5], 0.003, 1e-3)
names = ["top loads", "A pool fwd", "cls stage", "B logits", "C Hv", "D losses", "E+F gVt", "G pool bwd", "loss part"]
for it in range(3):
    eng.fused_step(); torch.cuda.synchronize()
    off, n = eng.plan.region("g_attn")
    t = eng.ws[off:off + 18].view(torch.int64).cpu().tolist()
    print("iter", it, " ".join(f"{nm}={(b - a)}" for nm, a, b in zip(names[1:], t[:-1], t[1:])), "total", t[8] - t[0])
build.build(force=True, verbose=False)
