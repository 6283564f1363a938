"""Step time of BASELINE configs[0]'s shape (hmdb_ucf_small: 5 classes, TemPooling/avgpool, source-only, 128+74 videos,
5 segments, 2048-d) on the GPU, both arithmetics.  usage: python tools/time_config1.py [steps]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ta3n_amd.engine import TrainEngine
from ta3n_amd.synthetic import synth_batch, synth_state
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 500
for bf16 in (False, True):
    eng = TrainEngine(128, 74, 5, 2048, 512, 5, dropout_i=0.5, dropout_v=0.5, clip=20.0, aggregation="avgpool", bf16=bf16, bf16_store=bf16)
    shapes = {n: s for n, _, s, _ in eng.plan.params}
    eng.load_state(synth_state(shapes, seed=7, scale="init"))
    xs, xt, ys, yt = synth_batch(5, 5, 2048, 128, 74, seed=1234)
    eng.set_batch(xs.cuda(), xt.cuda(), ys.cuda())
    for i in range(50):
        eng.train_step_pipelined([0, 0, 0], 0.0, 3e-2)
    eng.flush(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        eng.train_step_pipelined([0, 0, 0], 0.0, 3e-2)
    eng.flush(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ph = [(k, t, n, round(1e3 * ms, 2)) for k, t, n, ms in eng.time_phases(20)]
    print(f"config1 avgpool {'bf16' if bf16 else 'f32'}: {1e6 * dt:.1f} us/step -> {202 / dt:.0f} videos/s; per launch {ph}")
