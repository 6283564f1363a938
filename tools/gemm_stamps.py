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
eng = TrainEngine(CFG["Bs"], CFG["Bt"], CFG["T"], CFG["D"], CFG["F"], CFG["C"], phase_tiles=bench.DEFAULT_PHASE_TILES_BF16 if bf16 else bench.DEFAULT_PHASE_TILES,
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
    real = st[:, 5] > 0
    t0 = st[real, 0].min()
    end = st[real, 3].max() - t0
    life = st[real, 3] - st[real, 0]
    k = int(np.argmax(st[:, 3] * real))
    print(f"launch {i} tile {ph['tile']} tasks {n} (real {real.sum()}): span {end} ticks; start spread {st[real,0].max()-t0}; wg life avg {life.mean():.0f} max {life.max()}; "
          f"last finisher wg {k}: entry+{st[k,0]-t0} desc {st[k,1]-st[k,0]} kloop {st[k,2]-st[k,1]} epi {st[k,3]-st[k,2]} cost {st[k,4]} segs {st[k,5]}")
    # by cost class
    costs = np.unique(st[real, 4])
    for cst in costs[-4:]:
        m = real & (st[:, 4] == cst)
        print(f"    cost {cst}: {m.sum()} wgs, kloop avg {np.mean(st[m,2]-st[m,1]):.0f} max {np.max(st[m,2]-st[m,1])}, epi avg {np.mean(st[m,3]-st[m,2]):.0f}, desc avg {np.mean(st[m,1]-st[m,0]):.0f}, finish avg {np.mean(st[m,3])-t0:.0f}")
