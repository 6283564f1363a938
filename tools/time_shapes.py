"""Single-GPU step time at the other BASELINE shapes (configs[3]: Kinetics->Gameplay, 30 classes, 9 segments, 512+512 videos;
configs[4]: one 1024-d stream, 12 segments, 128+128 videos), both arithmetics, default (untuned) tile heuristics.
usage: python tools/time_shapes.py [steps]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ta3n_amd.engine import TrainEngine
from ta3n_amd.synthetic import synth_batch, synth_state
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for name, (Bs, Bt, T, D, C) in {"configs[3] T9 C30 b512+512": (512, 512, 9, 2048, 30), "configs[4] stream T12 D1024 b128+128": (128, 128, 12, 1024, 12)}.items():
    for bf16 in (False, True):
        eng = TrainEngine(Bs, Bt, T, D, 512, C, dropout_i=0.5, dropout_v=0.5, clip=20.0, bf16=bf16, bf16_store=bf16)
        shapes = {n: s for n, _, s, _ in eng.plan.params}
        eng.load_state(synth_state(shapes, seed=7, scale="init"))
        xs, xt, ys, yt = synth_batch(C, T, D, Bs, Bt, seed=1234)
        eng.set_batch(xs.cuda(), xt.cuda(), ys.cuda())
        for i in range(10):
            eng.train_step_pipelined([0.75, 0.75, 0.5], 0.003, 3e-2)
        eng.flush(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            eng.train_step_pipelined([0.75, 0.75, 0.5], 0.003, 3e-2)
        eng.flush(); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        twin = sum(1 for ph in eng.plan.description["phases"] if ph["kind"] == 0 and ph["group"] == 4 and ph["tile"] >= 16000)
        print(f"{name} {'bf16' if bf16 else 'f32'}: {1e6 * dt:.1f} us/step -> {(Bs + Bt) / dt:.0f} videos/s (twin launches {twin}, finite {bool(torch.isfinite(eng.P).all())})", flush=True)
        del eng
