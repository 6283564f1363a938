"""Gradient parity in full.  The golden-vector tests compare big tensors on 128 sampled entries + three
moments and, after the first update, with a loose absolute term; here EVERY element of EVERY gradient tensor of EVERY step is
compared with the CPU oracle (itself pinned to the reference by tests/test_oracle_golden.py), per tensor, in relative L2 and in
its largest deviation.  To keep the comparison about ONE step's arithmetic, the oracle is re-synchronised to the engine's
parameters before each step: no trajectory drift, so no ReLU unit flips between the two sides except on exact ties.
 * fp32 MFMA and the fp32-grade split arithmetic (f32x3): bounds F32_GRAD_* / F32X3_GRAD_* of ta3n_amd/tolerances.py, at the
   golden shapes, the headline shape and the full configs[3] / configs[4] shapes;
 * bf16: distance of logits AND gradients from the fp32 reference at the headline and configs[3] / configs[4] shapes
   (BF16_REF_*), next to the gate against the bf16-operand oracle in tests/test_gpu_bf16.py.
Measured floors are printed (pytest -s) and recorded in profiles/r03_parity_floors.txt."""
import numpy as np
import pytest
import torch

from golden_util import Golden, case_config, step_schedule
from oracle import ta3n_oracle as orc
from ta3n_amd import tolerances as tol
from ta3n_amd.engine import TrainEngine
from ta3n_amd.synthetic import synth_batch, synth_state

pytestmark = pytest.mark.gpu

SHAPES = {
    "config4_T9_C30_b512": dict(Bs=512, Bt=512, T=9, D=2048, F=512, C=30),
    "config5_T12_D1024": dict(Bs=128, Bt=128, T=12, D=1024, F=512, C=12),
    "headline": dict(Bs=128, Bt=74, T=5, D=2048, F=512, C=12),
    # (round 6) the heads kernel's pipelined relation loops with TWO videos per workgroup run for every batch above 224 videos
    # since round 5: more than 448 videos at 12 segments (11 relations: 5-6 per wave) and an odd count at 9, narrow features so the oracle stays fast
    "pipe_T12_b480": dict(Bs=256, Bt=224, T=12, D=256, F=128, C=12),
    "pipe_T9_b463": dict(Bs=232, Bt=231, T=9, D=256, F=128, C=30),
}


def _metrics(got, want):
    """{name: (rel. L2, max/scale, numel)} over two dicts of tensors."""
    out = {}
    for k, w in want.items():
        w = w.detach().double().cpu()
        g = got[k].detach().double().cpu().reshape(w.shape)
        d = g - w
        out[k] = ((d.pow(2).sum().sqrt() / (w.pow(2).sum().sqrt() + 1e-300)).item(), d.abs().max().item() / (w.abs().max().item() + 1e-300),
                  w.numel())
    return out


def _median(m):
    return float(np.median([v[0] for v in m.values()]))


def _check(grad_m, logit_err, arith):
    l2_tol, med_tol, mx_tol = ((tol.F32X3_GRAD_REL_L2, tol.F32X3_GRAD_REL_L2_MEDIAN, tol.F32X3_GRAD_MAX_SCALE) if arith.startswith("f32x3") else
                               (tol.F32_GRAD_REL_L2, tol.F32_GRAD_REL_L2_MEDIAN, tol.F32_GRAD_MAX_SCALE))
    over = []
    for s, m in enumerate(grad_m):
        for k, (l2, mx, n) in m.items():
            assert l2 <= l2_tol and mx <= mx_tol, f"step {s} {k}: rel. L2 {l2:.3e}, max/scale {mx:.3e}"
        if _median(m) > med_tol:
            over.append(s)
        for key, (err, rms) in logit_err[s].items():
            assert err < tol.LOGIT_ATOL, (s, key, err)
    # a ReLU tie that lands on the other side than in the oracle moves every upstream tensor (tolerances.py): split arithmetic only,
    # one step per case, bounded
    allowed = 1 if arith.startswith("f32x3") else 0
    assert len(over) <= allowed and all(_median(grad_m[s]) <= tol.F32X3_GRAD_REL_L2_MEDIAN_TIE for s in over), \
        "median rel. L2 per step: " + ", ".join(f"{_median(m):.3e}" for m in grad_m)


def _worst(m, n=3):
    return f"median L2 {_median(m):.1e}; worst " + ", ".join(f"{k} L2 {v[0]:.1e} max {v[1]:.1e}" for k, v in sorted(m.items(), key=lambda kv: -kv[1][0])[:n])


def _steps_against_resynced_oracle(shape, arith, steps, fused=True, wseed=11, xseed=21, lr=2e-3, wscale="trained", n_valid=None, clip=20.0,
                                   contract_m=None):
    Bs, Bt, T, D, Fc, Cn = (shape[k] for k in ("Bs", "Bt", "T", "D", "F", "C"))
    cfg = orc.Config(num_class=Cn, num_segments=T, feature_dim=D, fc_dim=Fc, dropout_i=0.0, dropout_v=0.0)
    params = synth_state(orc.param_shapes(cfg), seed=wseed, scale=wscale)
    kw = (dict(f32_split=True) if arith == "f32x3" else dict(f32_split=True, bf16_store=True) if arith == "f32x3p" else
          dict(bf16=True, bf16_store=True) if arith == "bf16" else {})      # f32x3p: the split arithmetic on stored hi / lo planes ("pair twins")
    eng = TrainEngine(Bs, Bt, T, D, Fc, Cn, dropout_i=0.0, dropout_v=0.0, clip=clip, fused=fused, **kw)
    eng.load_state(params)
    grad_m, logit_err = [], []
    for s in range(steps):
        xs, xt, ys, yt = synth_batch(Cn, T, D, Bs, Bt, seed=xseed + 7 * s)
        ns, nt = n_valid[s] if n_valid else (Bs, Bt)
        xs[ns:] = 0; xt[nt:] = 0
        # the oracle starts every step from the ENGINE's parameters and momentum
        state = orc.TrainState(params={k: v.detach().cpu().clone() for k, v in eng.param_views().items()}, lr=lr)
        state.momentum = {k: v.detach().cpu().clone() for k, v in eng.momentum_views().items()} if hasattr(state, "momentum") else None
        eng.set_batch(xs.cuda(), xt.cuda(), ys.cuda())
        eng.train_step([0.75, 0.75, 0.5], 0.003, lr, valid_source=ns, valid_target=nt, seed=s)
        torch.cuda.synchronize()
        if contract_m is not None:      # what the bf16 CONTRACT itself costs on these inputs: the oracle's bf16-operand mode against its fp32 mode
            st16 = orc.TrainState(params={k: v.clone() for k, v in state.params.items()}, lr=lr)
            st16.momentum = {k: v.clone() for k, v in state.momentum.items()}
            cfg16 = orc.Config(num_class=Cn, num_segments=T, feature_dim=D, fc_dim=Fc, dropout_i=0.0, dropout_v=0.0, arithmetic="bf16")
            res16 = orc.train_step(st16, xs, xt, ys, [0.75, 0.75, 0.5], 0.003, cfg16, clip=clip, n_src=ns, n_tgt=nt)
        res = orc.train_step(state, xs, xt, ys, [0.75, 0.75, 0.5], 0.003, cfg, clip=clip, n_src=ns, n_tgt=nt)
        got_g = {k: v for k, v in eng.param_views(eng.G).items() if k in res["grads"]}
        grad_m.append(_metrics(got_g, res["grads"]))
        if contract_m is not None:
            contract_m.append(_metrics(res16["grads"], res["grads"]))
        o = eng.outputs()
        errs = {}
        for key, pick in (("out", lambda r: r["out"]), ("pred_rel", lambda r: r["pred_domain"][0]), ("pred_vid", lambda r: r["pred_domain"][1]),
                          ("pred_frm", lambda r: r["pred_domain"][2])):
            want = torch.cat((pick(res["src"]), pick(res["tgt"])), 0).detach()
            errs[key] = ((o[key].cpu().reshape(want.shape) - want).abs().max().item(), want.pow(2).mean().sqrt().item())
        logit_err.append(errs)
    return grad_m, logit_err


@pytest.mark.parametrize("name,arith,fused", [(n, a, f) for n in ("tiny_T5", "tiny_T9", "mid_T12", "headline") for a in ("f32", "f32x3", "f32x3p")
                                              for f in (True, False)
                                              if f or (n in ("tiny_T5", "headline") and a != "f32x3p")])      # unfused launch lists on two cases; pair
def test_every_gradient_element_matches_the_oracle(name, arith, fused, capsys):                              # twins are read by the fused step only
    g = Golden(name)
    c = case_config(g)
    shape = dict(Bs=c["Bs"], Bt=c["Bt"], T=c["T"], D=c["D"], F=c["fc_dim"], C=c["C"])
    sched = step_schedule(c)
    grad_m, logit_err = _steps_against_resynced_oracle(shape, arith, steps=max(3, len(sched)), fused=fused, wseed=c["wseed"], xseed=c["xseed"],
                                                       lr=c["lr"], wscale=c["wscale"], clip=c["clip"],
                                                       n_valid=[(st["n_src"], st["n_tgt"]) for st in sched] + [(c["Bs"], c["Bt"])] * 3)
    with capsys.disabled():
        for s, m in enumerate(grad_m):
            print(f"\n[{arith} grads vs oracle] {name} fused={fused} step {s}: {_worst(m)} | logits max {max(e[0] for e in logit_err[s].values()):.1e}")
    _check(grad_m, logit_err, arith)


@pytest.mark.parametrize("shape,arith", [(sh, a) for sh in ("config4_T9_C30_b512", "config5_T12_D1024") for a in ("f32", "f32x3", "f32x3p")] +
                         [("pipe_T12_b480", "f32"), ("pipe_T9_b463", "f32")])
def test_full_shape_gradients_match_the_oracle(shape, arith, capsys):
    """BASELINE configs[3] / configs[4] at full size, two steps, all gradients in full, fp32 MFMA and the split arithmetic."""
    grad_m, logit_err = _steps_against_resynced_oracle(SHAPES[shape], arith, steps=2)
    with capsys.disabled():
        for s, m in enumerate(grad_m):
            print(f"\n[{arith} grads vs oracle] {shape} step {s}: {_worst(m)} | logits max {max(e[0] for e in logit_err[s].values()):.1e}")
    _check(grad_m, logit_err, arith)


@pytest.mark.parametrize("shape", ["headline", "config4_T9_C30_b512", "config5_T12_D1024"])      # (the benchmarked shapes: the bounds are their measured floors)
def test_bf16_distance_from_the_fp32_reference_logits_and_gradients(shape, capsys):
    """What rounding the contraction operands to bf16 costs against the REFERENCE's fp32 arithmetic, for the logits and for every
    gradient tensor, at the benchmarked shapes (the gate against the bf16-operand oracle is tests/test_gpu_bf16.py)."""
    contract_m = []
    grad_m, logit_err = _steps_against_resynced_oracle(SHAPES[shape], "bf16", steps=2, contract_m=contract_m)
    with capsys.disabled():
        for s, m in enumerate(grad_m):
            big = {k: v for k, v in m.items() if v[2] >= 4096}
            ratio = max(v[0] / (tol.BF16_REF_GRAD_CONTRACT_FACTOR * contract_m[s][k][0] + tol.BF16_REF_GRAD_FLOOR) for k, v in big.items())
            print(f"\n[bf16 vs fp32 reference] {shape} step {s}: {_worst(big)} | median L2 {np.median([v[0] for v in m.values()]):.1e} | "
                  f"logits max/rms {max(e[0] / (e[1] + 1e-30) for e in logit_err[s].values()):.1e} | the contract itself (oracle bf16 mode vs fp32 mode): "
                  f"{_worst({k: v for k, v in contract_m[s].items() if v[2] >= 4096})} | worst fraction of the per-tensor bound {ratio:.2f}")
    for s, m in enumerate(grad_m):
        for k, (l2, mx, n) in m.items():
            if n >= 4096:
                # no further from the reference than a small multiple of what the arithmetic contract itself costs on these very inputs
                # (profiles/r04_bf16_gradient_deviation_attribution.txt: that cost is the ReLU on/off pattern - 0.08 % of the hidden units
                # land on the other side of zero - not the rounding of the products), and never beyond the absolute cap
                bound = min(tol.BF16_REF_GRAD_REL_L2, tol.BF16_REF_GRAD_CONTRACT_FACTOR * contract_m[s][k][0] + tol.BF16_REF_GRAD_FLOOR)
                assert l2 <= bound, f"step {s} {k}: rel. L2 {l2:.3e} > {bound:.3e} (contract: {contract_m[s][k][0]:.3e})"
        assert np.median([v[0] for v in m.values()]) <= tol.BF16_REF_GRAD_REL_L2_MEDIAN
        for key, (err, rms) in logit_err[s].items():
            assert err <= tol.BF16_REF_LOGIT_REL_RMS * rms + 1e-7, (s, key, err, rms)
