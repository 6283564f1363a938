"""Mask-synchronised gradient parity.  tests/test_gpu_gradients.py compares every gradient element with the
free-running oracle and needs a 5e-3 per-tensor bound: a hidden unit within round-off of zero that lands on the other side in the
two computations moves a first-layer discriminator gradient by ~1/rows of its norm - an indexing bug of that size would hide under
"ReLU flip".  Here the oracle is forced to the ENGINE's on/off patterns (read back from the workspace: F1, Hf, the TRN tuple
activations Zr, the relation / video discriminator hidden layers Hr, Hv; oracle/ta3n_oracle.py: _relu_m), so what is left is the
arithmetic of one step: fp32 summation order.  Every element of every gradient tensor, fp32 MFMA, per-tensor relative L2 <=
F32_MASKED_GRAD_REL_L2 (ta3n_amd/tolerances.py) - 25 x tighter than the free-running bound, the median over a step's tensors 2e-5; the free-running test stays beside it.
Also checked on the way: the hidden activations themselves against the oracle's (they are not outputs of VideoModel.forward, so no
other test sees them)."""
import numpy as np
import pytest
import torch

from oracle import ta3n_oracle as orc
from ta3n_amd import tolerances as tol
from ta3n_amd.engine import TrainEngine
from ta3n_amd.synthetic import synth_batch, synth_state

pytestmark = pytest.mark.gpu

SHAPES = {
    "tiny_T5": dict(Bs=6, Bt=4, T=5, D=512, F=64, C=12),
    "ragged_T3": dict(Bs=40, Bt=30, T=3, D=256, F=128, C=7),
    "headline": dict(Bs=128, Bt=74, T=5, D=2048, F=512, C=12),
    "config5_T12_D1024": dict(Bs=128, Bt=128, T=12, D=1024, F=512, C=12),
}


def _engine_masks(eng, n_tuples):
    B, T, F, NR = eng.B, eng.T, eng.F, eng.T - 1
    full = dict(F1=eng.region("F1", (B * T, F)), Hf=eng.region("Hf", (B * T, F)), Z=eng.region("Zr", (B, n_tuples, 256)),
                Hr=eng.region("Hr", (B, NR, 256)), Hv=eng.region("Hv", (B, 256)))
    full = {k: v.detach().cpu() for k, v in full.items()}
    rows = lambda k, lo, hi: full[k][lo * T:hi * T] if k in ("F1", "Hf") else full[k][lo:hi]
    act = tuple({k: rows(k, lo, hi) for k in full} for lo, hi in ((0, eng.Bs), (eng.Bs, B)))
    return tuple({k: v > 0 for k, v in a.items()} for a in act), act


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "unfused"])
@pytest.mark.parametrize("name", sorted(SHAPES))
def test_every_gradient_element_against_the_mask_synchronised_oracle(name, fused, capsys):
    sh = SHAPES[name]
    Bs, Bt, T, D, Fc, Cn = (sh[k] for k in ("Bs", "Bt", "T", "D", "F", "C"))
    if name != "tiny_T5" and not fused:
        pytest.skip("the unfused launch lists are covered at the small shape")
    cfg = orc.Config(num_class=Cn, num_segments=T, feature_dim=D, fc_dim=Fc, dropout_i=0.0, dropout_v=0.0)
    params = synth_state(orc.param_shapes(cfg), seed=11, scale="trained")
    eng = TrainEngine(Bs, Bt, T, D, Fc, Cn, dropout_i=0.0, dropout_v=0.0, clip=20.0, fused=fused)
    eng.load_state(params)
    n_tuples = sum(len(s) for s in orc.selected_relations(T))
    worst, lines = 0.0, []
    for s in range(2):
        xs, xt, ys, yt = synth_batch(Cn, T, D, Bs, Bt, seed=21 + 7 * s)
        ns, nt = (Bs, Bt) if (s == 0 or name != "ragged_T3") else (Bs - 3, Bt - 5)      # second step of the ragged case: dummy rows
        xs[ns:] = 0; xt[nt:] = 0
        state = orc.TrainState(params={k: v.detach().cpu().clone() for k, v in eng.param_views().items()}, lr=2e-3)
        state.momentum = {k: v.detach().cpu().clone() for k, v in eng.momentum_views().items()}
        eng.set_batch(xs.cuda(), xt.cuda(), ys.cuda())
        eng.train_step([0.75, 0.75, 0.5], 0.003, 2e-3, valid_source=ns, valid_target=nt, seed=s)
        torch.cuda.synchronize()
        masks, act = _engine_masks(eng, n_tuples)
        res = orc.train_step(state, xs, xt, ys, [0.75, 0.75, 0.5], 0.003, cfg, clip=20.0, n_src=ns, n_tgt=nt, masks=masks)
        # the hidden activations themselves (valid rows; the oracle computes x * mask, so where the engine's unit is on and the oracle's
        # pre-activation is a round-off below zero the oracle holds that tiny negative number: compared at the bound of the outputs)
        for d, (dom, nv) in enumerate((("src", ns), ("tgt", nt))):
            for k, a in act[d].items():
                want = res[dom]["hidden"][k].detach()
                rows = nv * T if k in ("F1", "Hf") else nv
                err = (a[:rows] - want[:rows]).abs().max().item()
                assert err <= 2e-4 * max(1.0, want[:rows].abs().max().item()), (name, s, dom, k, err)
        got = {k: v.detach().cpu() for k, v in eng.param_views(eng.G).items() if k in res["grads"]}
        per = {}
        for k, w in res["grads"].items():
            dlt = (got[k].double().reshape(w.shape) - w.double())
            per[k] = (dlt.pow(2).sum().sqrt() / (w.double().pow(2).sum().sqrt() + 1e-300)).item()
        med = float(np.median(list(per.values())))
        top = sorted(per.items(), key=lambda kv: -kv[1])[:3]
        lines.append(f"[masked fp32] {name} {'fused' if fused else 'unfused'} step {s}: median rel. L2 {med:.2e}; worst " +
                     ", ".join(f"{k} {v:.2e}" for k, v in top))
        worst = max(worst, top[0][1])
        for k, v in per.items():
            assert v <= tol.F32_MASKED_GRAD_REL_L2, (name, s, k, v)
        assert med <= tol.F32_MASKED_GRAD_REL_L2_MEDIAN, (name, s, med)
    with capsys.disabled():
        print("\n" + "\n".join(lines))
