"""Debug aid (needs a library built with -DTA3N_GEMM_STAMPS in a directory of its own:
    TA3N_LIBDIR=$PWD/ta3n_amd/lib_stamps TA3N_EXTRA_FLAGS=-DTA3N_GEMM_STAMPS python -m ta3n_amd.build
    TA3N_LIBDIR=$PWD/ta3n_amd/lib_stamps python tools/gemm_stamps.py [bf16|f32] [config 2|4|5]
): per-workgroup s_memtime stamps of every GEMM launch of the fused step - entry / descriptors loaded / K loop done / epilogue done,
and inside the K loop the time thread 0 spends WAITING for a stage (s_waitcnt + s_barrier) against the time it spends issuing the next
stage's DMA, LDS reads and MFMAs - printed per launch and per tile length: where a launch's time goes."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ta3n_amd import _lib
from ta3n_amd.engine import TrainEngine
from ta3n_amd.synthetic import synth_batch, synth_state
sys.path.insert(0, ROOT)
import bench
CFG = bench.CONFIGS[int(sys.argv[2]) if len(sys.argv) > 2 else 2]["shape"]
bf16 = (sys.argv[1] if len(sys.argv) > 1 else "bf16") == "bf16"
print(f"## {'bf16 (twins)' if bf16 else 'fp32'} shape {CFG}")
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
n_ph = L.ta3n_num_phases(eng.plan.handle, 4)
phases = [ph for ph in eng.plan.description["phases"] if ph["group"] == 4]
args = (eng.plan.handle, eng.X.data_ptr(), eng.P.data_ptr(), eng.G.data_ptr(), eng.ws.data_ptr())
SL = 16
stamps = eng.region("stamps").view(torch.int64)
for i, ph in enumerate(phases):
    if ph["kind"] != 0:
        L.ta3n_train_step_range(*args, i, 1, eng._stream()); continue
    torch.cuda.synchronize()
    L.ta3n_train_step_range(*args, i, 1, eng._stream())     # warm: code and operands in cache, as inside a running step
    torch.cuda.synchronize()
    stamps.zero_()
    L.ta3n_train_step_range(*args, i, 1, eng._stream())
    torch.cuda.synchronize()
    n = min(ph["task_count"], 8192)
    st = stamps[: n * SL].cpu().numpy().reshape(n, SL).astype(np.int64)
    real = st[:, 7] > 0
    if not real.any():
        print(f"launch {i}: no stamps"); continue
    t0 = st[real, 0].min()
    TICK = 1.0 / 2400      # s_memtime counts shader clocks; us at the nominal 2.4 GHz (a launch timed with HIP events calibrates it: span ~ duration)
    print(f"launch {i} tile {ph['tile']} tasks {n} (real {real.sum()}): span {(st[real, 5].max() - t0) * TICK:.2f} us, last entry {(st[real, 0].max() - t0) * TICK:.2f} us after the first")
    for cst in np.unique(st[real, 6])[-5:]:
        m = real & (st[:, 6] == cst)
        d = lambda a, b: np.mean(st[m, a] - st[m, b]) * TICK
        ns = np.maximum(st[m, 10], 1)
        print(f"    K {cst:5d} x{m.sum():4d}: desc {d(1, 0):5.2f}  kloop {d(2, 1):6.2f} (thread 0: waiting for stage+barrier {np.mean(st[m, 8]) * TICK:6.2f}, issue+compute {np.mean(st[m, 9]) * TICK:6.2f}, "
              f"{np.mean(ns):.1f} stages -> {np.mean(st[m, 8] / ns) * TICK * 1e3:5.0f} + {np.mean(st[m, 9] / ns) * TICK * 1e3:5.0f} ns per stage)  wait-waves {d(3, 2):5.2f}  to-lds {d(4, 3):5.2f}  "
              f"combine+store {d(5, 4):5.2f}  | finish {np.mean(st[m, 5] - t0) * TICK:6.2f} us")
