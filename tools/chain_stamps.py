"""Debug aid for chained launches (needs the library built with -DTA3N_GEMM_STAMPS=2: python tools/chain_stamps.py build):
per-workgroup s_memtime stamps of the chained forward / backward launch of the fused bf16 step - workgroup entry, end of the
wait, descriptors loaded, K loop done, epilogue done, exit - summarised per class of task (producer / consumer / neither) as a
timeline relative to the first workgroup's entry.  usage: python tools/chain_stamps.py [bf16|f32] (env TA3N_CHAIN_* knobs apply)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) > 1 and sys.argv[1] == "build":
    from ta3n_amd import build
    build.build(force=True, extra_flags=("-DTA3N_GEMM_STAMPS=2",))
    sys.exit(0)
import numpy as np, torch
from ta3n_amd import _lib
from ta3n_amd.engine import TrainEngine
from ta3n_amd.synthetic import synth_batch, synth_state
import bench, plan_interp
CFG = bench.CFG
bf16 = (sys.argv[1] if len(sys.argv) > 1 else "bf16") == "bf16"
chain = os.environ.get("TA3N_CHAIN", "1") == "1"
eng = TrainEngine(CFG["Bs"], CFG["Bt"], CFG["T"], CFG["D"], CFG["F"], CFG["C"], bf16=bf16, bf16_store=bf16, chain=chain)
eng.load_state(synth_state({n: s for n, _, s, _ in eng.plan.params}, seed=7, scale="init"))
xs, xt, ys, yt = synth_batch(CFG["C"], CFG["T"], CFG["D"], CFG["Bs"], CFG["Bt"], seed=1234)
eng.set_batch(xs.cuda(), xt.cuda(), ys.cuda())
eng.set_hyper([0.75, 0.75, 0.5], 0.003, 0.03)
for _ in range(3):
    eng.fused_step()
torch.cuda.synchronize()
L = _lib.lib()
L.ta3n_debug_stamps.argtypes = [C.c_void_p, C.c_int]
segs, tasks, phases_c, geom, tup, tf = plan_interp.plan_arrays(eng.plan)
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
    tk = [tasks[ph["task_begin"] + k] for k in range(n)]
    real = np.array([t.seg_count > 0 for t in tk])
    # s_memtime counters are per XCD (workgroup b runs on XCD b % 8): times are taken relative to the first entry on the same XCD
    xcd = np.arange(n) % 8
    t0x = np.array([st[real & (xcd == x), 6].min() for x in range(8)])
    st[:, :8] -= t0x[xcd][:, None]
    t0 = 0
    us = lambda v: v / 2100.0      # s_memtime ticks = shader cycles (~2.1 GHz under load)
    print(f"launch {i} tile {ph['tile']} tasks {n} chain_counters {ph['chain_counters']}: span {us(st[real, 7].max() - t0):.1f} us")
    cls = np.array([("consumer+producer" if (t.wait_count > 0 and t.sig >= 0) else "consumer" if t.wait_count > 0 else "producer" if t.sig >= 0 else "plain") for t in tk])
    for c in ("producer", "consumer+producer", "consumer", "plain"):
        m = real & (cls == c)
        if not m.any():
            continue
        q = lambda a: f"{us(np.percentile(st[m, a] - t0, 5)):6.1f}/{us(np.median(st[m, a] - t0)):6.1f}/{us(np.percentile(st[m, a] - t0, 95)):6.1f}"
        d = lambda a, b: f"{us(np.median(st[m, a] - st[m, b])):5.1f}"
        print(f"   {c:18s} x{m.sum():4d}  entry {q(6)}  wait-end {q(0)}  kloop-end {q(2)}  end {q(5)}  exit {q(7)} us (5%/median/95%) | "
              f"waited {d(0, 6)}  desc {d(1, 0)}  kloop {d(2, 1)}  epilogue {d(5, 2)}  exit {d(7, 5)}")
