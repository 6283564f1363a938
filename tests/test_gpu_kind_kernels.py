"""Kind-specialised GEMM kernels (gemm_tiles<..., KV>: only the K-loop variants and epilogue paths a launch's tasks use - a fraction
of the code to fetch when the kernel changes between launches; picked by launch_gemm from ta3n_plan::phase_kinds) are the same
arithmetic as the full kernels: bit-identical parameters, gradients and losses after pipelined steps at the headline shape, in the
bf16-twin and the pair-twin arithmetic.  (TA3N_KIND_KERNELS is read once per process: two subprocesses.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

SCRIPT = r"""
import hashlib, sys, torch
sys.path.insert(0, "tests")
from ta3n_amd.engine import TrainEngine
from ta3n_amd.synthetic import synth_batch, synth_state
kw = dict(bf16=True, bf16_store=True) if sys.argv[1] == "bf16" else dict(f32_split=True, bf16_store=True)
eng = TrainEngine(128, 74, 5, 2048, 512, 12, dropout_i=0.5, dropout_v=0.5, **kw)
shapes = {n: s for n, _, s, _ in eng.plan.params}
eng.load_state(synth_state(shapes, seed=7, scale="trained"))
for step in range(4):
    xs, xt, ys, yt = synth_batch(12, 5, 2048, 128, 74, seed=50 + step)
    eng.set_batch(xs.cuda(), xt.cuda(), ys.cuda())
    eng.train_step_pipelined([0.75, 0.75, 0.5], 0.003, 0.01)
eng.flush(); torch.cuda.synchronize()
h = hashlib.sha256()
for t in (eng.P, eng.G, eng.M, eng.region("losses")[:6]):
    h.update(t.cpu().numpy().tobytes())
print("HASH", h.hexdigest(), float(eng.region("losses")[0]))
"""


@pytest.mark.parametrize("arith", ["bf16", "f32x3p"])
def test_specialised_kernels_are_bit_identical_to_the_full_ones(arith):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for flag in ("0", "1"):
        env = dict(os.environ, TA3N_KIND_KERNELS=flag)
        r = subprocess.run([sys.executable, "-c", SCRIPT, arith], cwd=root, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        line = [l for l in r.stdout.splitlines() if l.startswith("HASH")][-1].split()
        out[flag] = line[1]
        assert float(line[2]) == float(line[2]) and float(line[2]) > 0      # a finite loss
    assert out["0"] == out["1"]
