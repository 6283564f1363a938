"""Stage timeline of the fused heads kernel (workgroup 0): build the library with -DTA3N_HEADS_TIMING, run one
fused step, print s_memtime deltas between stage boundaries.  Debug aid, not part of the product.
usage: python tools/heads_timing.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ta3n_amd import _lib
# a library whose ta3n_heads.o was compiled with -DTA3N_HEADS_TIMING (build it where hipcc is, it travels with the tree):
#   hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTA3N_HEADS_TIMING -x hip -c ta3n_amd/csrc/ta3n_heads.hip -o tools/lib_timing/ta3n_heads.o
#   hipcc -shared -fPIC --offload-arch=gfx950 -o tools/lib_timing/libta3n_hip.so <the other objects of ta3n_amd/lib> tools/lib_timing/ta3n_heads.o -ldl
# (round 4: TA3N_LIBDIR=ta3n_amd/lib_ab TA3N_EXTRA_FLAGS=-DTA3N_HEADS_TIMING python -m ta3n_amd.build, then run with the same TA3N_LIBDIR)
if not os.environ.get("TA3N_LIBDIR"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib_timing", "libta3n_hip.so")
from ta3n_amd.engine import TrainEngine
BF16 = "--bf16" in sys.argv
shape = [int(v) for v in sys.argv[1:7]] if len(sys.argv) >= 7 and sys.argv[1].isdigit() else [128, 74, 5, 2048, 512, 12]
print("shape", shape, "bf16" if BF16 else "f32")
eng = TrainEngine(*shape, bf16=BF16, bf16_store=BF16)
eng.X.uniform_(0, 1)
for v in eng.param_views().values(): v.normal_(0, 0.02)
eng.set_hyper([0.75, 0.75, 0.5], 0.003, 1e-3)
names = ["top loads", "A pool fwd", "cls stage", "B logits", "C Hv", "D losses", "E+F gVt", "G pool bwd", "loss part"]
for it in range(3):
    eng.fused_step(); torch.cuda.synchronize()
    off, n = eng.plan.region("g_attn")
    t = eng.ws[off:off + 18].view(torch.int64).cpu().tolist()
    print("iter", it, " ".join(f"{nm}={(b - a)}" for nm, a, b in zip(names[1:], t[:-1], t[1:])), "total", t[8] - t[0], "(s_memtime ticks)")
