"""Debug aid (needs the library built with -DTA3N_GEMM_STAMPS: python tools/gemm_stamps.py build): per-workgroup s_memtime
stamps of every GEMM launch of the fused bf16 step - entry / descriptors loaded / K loop done / epilogue done - printed as a
summary per launch: how long the launch runs, how long its longest workgroup lives and where that time goes."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "build":
    from ta3n_amd import build
    build.build(force=True, extra_flags=("-DTA3N_GEMM_STAMPS",))
    sys.exit(0)
import numpy as np, torch
from ta3n_amd import _lib
from ta3n_amd.engine import TrainEngine
from ta3n_amd.synthetic import synth_batch, synth_state
sys.path.insert(0, ROOT)
import bench
CFG = bench.CFG
bf16 = (sys.argv[1] if len(sys.argv) > 1 else "bf16") == "bf16"
eng = TrainEngine(CFG["Bs"], CFG["Bt"], CFG["T"], CFG["D"], CFG["F"], CFG["C"], 
                  bf16=bf16, bf16_store=bf16)
eng.load_state(synth_state({n: s for n, _, s, _ in eng.plan.params}, seed=7, scale="init"))
xs, xt, ys, yt = synth_batch(CFG["C"], CFG["T"], CFG["D"], CFG["Bs"], CFG["Bt"], seed=1234)
eng.set_batch(xs.cuda(), xt.cuda(), ys.cuda())
eng.set_hyper([0.75, 0.75, 0.5], 0.003, 0.03)
for _ in range(3):
    eng.fused_step()
torch.cuda.synchronize()
L = _lib.lib()
L.ta3n_debug_stamps.argtypes = [C.c_void_p, C.c_int]
n_ph = L.ta3n_num_phases(eng.plan.handle, 4)
phases = [ph for ph in eng.plan.description["phases"] if ph["group"] == 4]
args = (eng.plan.handle, eng.X.data_ptr(), eng.P.data_ptr(), eng.G.data_ptr(), eng.ws.data_ptr())
for i, ph in enumerate(phases):
    if ph["kind"] != 0:
        L.ta3n_train_step_range(*args, i, 1, eng._stream()); continue
    torch.cuda.synchronize()
    L.ta3n_train_step_range(*args, i, 1, eng._stream())
    torch.cuda.synchronize()
    n = ph["task_count"]
    buf = np.zeros(n * 8, np.uint64)
    L.ta3n_debug_stamps(buf.ctypes.data, n * 8)
    st = buf.reshape(n, 8).astype(np.int64)
    real = st[:, 7] > 0
    t0 = st[real, 0].min()
    print(f"launch {i} tile {ph['tile']} tasks {n} (real {real.sum()}): span {st[real, 5].max() - t0} ticks, entry spread {st[real, 0].max() - t0}")
    for cst in np.unique(st[real, 6])[-4:]:
        m = real & (st[:, 6] == cst)
        d = lambda a, b: np.mean(st[m, a] - st[m, b])
        print(f"    cost {cst:5d} x{m.sum():4d}: desc {d(1, 0):6.0f}  kloop {d(2, 1):6.0f}  wait-waves {d(3, 2):5.0f}  lds-transpose {d(4, 3):5.0f}  combine+store {d(5, 4):5.0f}  | finish {np.mean(st[m, 5]) - t0:7.0f}")
