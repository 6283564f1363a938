"""Does the bf16 arithmetic (the headline number's) TRAIN like the reference's fp32?  (nothing ran more than
three steps in bf16 against the fp32 reference.)

A learnable synthetic domain-adaptation task (ta3n_amd/synthetic.py: task_batch - class-dependent feature means, a shifted and
rescaled target domain) is trained for 300 steps of main.train's loop (main.py:348-621: DANN beta and learning-rate schedules, clip 20,
Nesterov SGD, a fresh batch every step) from identical parameters by
  * the CPU oracle in fp32 (oracle/ta3n_oracle.py, pinned to the reference by tests/test_oracle_golden.py) - the trajectory of the
    reference's own arithmetic,
  * the HIP engine in fp32 (fp32 MFMA), in the fp32-grade split arithmetic (f32x3 on stored hi / lo planes) and in bf16 (twins).
Trajectories of a ReLU network diverge element-wise whatever the arithmetic (a unit within round-off of zero lands on the other side),
so what is compared is what training is judged by: the loss curves (window means) and the accuracy on held-out videos of both domains.
Bounds in ta3n_amd/tolerances.py (TRAIN_*); the measured values are written to gpurun_out/ for profiles/.
Dropout is off on this leg (the oracle and the kernels draw different masks); the second test trains the bf16 and the fp32 engine WITH
dropout 0.5 / 0.5 on identical masks (the mask of an element is a hash of (seed, index): the same in every arithmetic)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import ta3n_oracle as orc
from ta3n_amd import tolerances as tol
from ta3n_amd.engine import TrainEngine, beta_dann, lr_dann
from ta3n_amd.synthetic import synth_state, task_batch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SH = dict(Bs=32, Bt=24, T=5, D=256, F=64, C=5)
STEPS, WINDOW, LR0, GAMMA = 300, 25, 0.01, 0.003
EARLY = 20                      # steps over which the trajectories are compared step by step (before ReLU flips decorrelate them)
KW = {"f32": {}, "f32x3": dict(f32_split=True, bf16_store=True), "bf16": dict(bf16=True, bf16_store=True)}


def _schedule(i):
    p = i / STEPS
    b = beta_dann(p)
    return [b, b, b], GAMMA, (LR0 if i == 0 else lr_dann(LR0, p))


def _windows(v):
    v = np.asarray(v, dtype=np.float64)
    return v[: len(v) // WINDOW * WINDOW].reshape(-1, WINDOW).mean(1)


def _heldout():
    return [task_batch(SH["C"], SH["T"], SH["D"], 100, 100, step=10_000 + k, task_seed=3) for k in range(2)]


def _engine_run(arith, dropout, F=None):
    F = SH["F"] if F is None else F
    eng = TrainEngine(SH["Bs"], SH["Bt"], SH["T"], SH["D"], F, SH["C"], dropout_i=dropout, dropout_v=dropout, clip=20.0, **KW[arith])
    eng.load_state(synth_state({n: s for n, _, s, _ in eng.plan.params}, seed=5, scale="trained"))
    curve = {"loss": [], "loss_c": [], "loss_a": [], "loss_e": []}
    for i in range(STEPS):
        xs, xt, ys, _ = task_batch(SH["C"], SH["T"], SH["D"], SH["Bs"], SH["Bt"], step=i, task_seed=3)
        eng.set_batch(xs.cuda(), xt.cuda(), ys.cuda())
        eng.train_step(*_schedule(i), seed=i)
        l = eng.losses()
        curve["loss"].append(l["loss"]); curve["loss_c"].append(l["loss_c"]); curve["loss_e"].append(l["loss_e"])
        curve["loss_a"].append(l["loss_adv_rel"] + l["loss_adv_vid"] + l["loss_adv_frm"])
    # held-out accuracy, both domains (main.validate's bookkeeping on the device; at most batch_source videos per call)
    acc = []
    ev = TrainEngine(100, 1, SH["T"], SH["D"], F, SH["C"], dropout_i=0.0, dropout_v=0.0, **KW[arith])
    ev.load_state(eng.state_dict())
    for dom in (0, 1):
        first = True
        for xs, xt, ys, yt in _heldout():
            ev.evaluate_batch((xs, xt)[dom].cuda(), (ys, yt)[dom].cuda(), reset=first)
            first = False
        acc.append(ev.eval_results()["prec1"])
    assert bool(torch.isfinite(eng.P).all())
    return curve, acc


def _oracle_run():
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    cfg = orc.Config(num_class=SH["C"], num_segments=SH["T"], feature_dim=SH["D"], fc_dim=SH["F"], dropout_i=0.0, dropout_v=0.0)
    state = orc.TrainState(params=synth_state(orc.param_shapes(cfg), seed=5, scale="trained"), lr=LR0)
    curve = {"loss": [], "loss_c": [], "loss_a": [], "loss_e": []}
    for i in range(STEPS):
        xs, xt, ys, _ = task_batch(SH["C"], SH["T"], SH["D"], SH["Bs"], SH["Bt"], step=i, task_seed=3)
        beta, gamma, lr = _schedule(i)
        state.lr = lr
        res = orc.train_step(state, xs, xt, ys, beta, gamma, cfg)
        pt = {k: float(v) for k, v in res["parts"].items()}
        curve["loss"].append(float(res["loss"])); curve["loss_c"].append(pt["loss_c"]); curve["loss_e"].append(pt["loss_e"])
        curve["loss_a"].append(pt["loss_a"])
    acc = []
    with torch.no_grad():
        for dom in (0, 1):
            hit = n = 0
            for xs, xt, ys, yt in _heldout():
                out = orc.forward_domain(state.params, (xs, xt)[dom], [0.0, 0.0, 0.0], cfg)["out"]
                hit += int((out.argmax(1) == (ys, yt)[dom]).sum()); n += out.shape[0]
            acc.append(100.0 * hit / n)
    return curve, acc


def _curve_distance(a, b):
    """`early`: largest per-step relative difference of the total loss over the first EARLY steps (the trajectories still coincide up
    to the arithmetic's rounding); `late_*`: the MEDIANS of the logged quantities over the last 100 steps (adversarial training is
    spiky and the spikes of two decorrelated trajectories do not line up: means and windows are dominated by them) - relative
    difference for the total and the adversarial loss (both O(2)), the larger of the two values for the classification loss and the
    attentive entropy (both -> 0 on a learnt task)"""
    la, lb = np.asarray(a["loss"][:EARLY]), np.asarray(b["loss"][:EARLY])
    med = lambda c, k: float(np.median(c[k][-100:]))
    return {"early": float(np.max(np.abs(la - lb) / np.abs(lb))),
            "late_loss_rel": abs(med(a, "loss") - med(b, "loss")) / med(b, "loss"),
            "late_loss_a_rel": abs(med(a, "loss_a") - med(b, "loss_a")) / med(b, "loss_a"),
            "late_loss_c_max": max(med(a, "loss_c"), med(b, "loss_c")),
            "late_loss_e_max": max(med(a, "loss_e"), med(b, "loss_e"))}


def _assert_tracks(dist, arith, acc, ref_acc):
    assert dist["early"] <= (tol.TRAIN_EARLY_REL_BF16 if arith == "bf16" else tol.TRAIN_EARLY_REL_F32), (arith, dist)
    assert dist["late_loss_rel"] <= tol.TRAIN_LATE_REL and dist["late_loss_a_rel"] <= tol.TRAIN_LATE_REL, (arith, dist)
    assert dist["late_loss_c_max"] <= tol.TRAIN_LATE_LOSS_C and dist["late_loss_e_max"] <= tol.TRAIN_LATE_LOSS_E, (arith, dist)
    assert all(abs(x - y) <= tol.TRAIN_ACC_POINTS for x, y in zip(acc, ref_acc)), (arith, acc, ref_acc)


def test_300_steps_track_the_fp32_oracle_trajectory():
    ref_curve, ref_acc = _oracle_run()
    assert _windows(ref_curve["loss_c"])[-1] < 0.25 * _windows(ref_curve["loss_c"])[0], "the task must be learnable"
    assert min(ref_acc) > 90.0, ref_acc
    report = {"oracle_fp32": {"acc_source_target": ref_acc, "loss_c_windows": _windows(ref_curve["loss_c"]).round(4).tolist()}}
    for arith in ("f32", "f32x3", "bf16"):
        curve, acc = _engine_run(arith, 0.0)
        report[arith] = {"acc_source_target": acc, "distance_from_oracle": _curve_distance(curve, ref_curve),
                         "loss_c_windows": _windows(curve["loss_c"]).round(4).tolist()}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "training_equivalence_300_steps.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report))
    for arith in ("f32", "f32x3", "bf16"):
        _assert_tracks(report[arith]["distance_from_oracle"], arith, report[arith]["acc_source_target"], ref_acc)


def test_300_steps_with_dropout_bf16_tracks_the_fp32_engine():
    """The reference's configuration trains with dropout 0.5 / 0.5: bf16 against the fp32 MFMA engine on identical dropout masks, with a
    128-wide shared layer (at 64 channels half of them dropped leave the task unlearnt in 300 steps - in the fp32 oracle too)."""
    ref_curve, ref_acc = _engine_run("f32", 0.5, F=128)
    curve, acc = _engine_run("bf16", 0.5, F=128)
    dist = _curve_distance(curve, ref_curve)
    med = lambda c, k: float(np.median(c[k][-100:]))
    late = {k: (med(curve, k), med(ref_curve, k)) for k in ("loss_c", "loss_e")}
    report = {"f32_engine": {"acc_source_target": ref_acc, "loss_c_windows": _windows(ref_curve["loss_c"]).round(4).tolist()},
              "bf16": {"acc_source_target": acc, "distance_from_f32_engine": dist, "late_medians_bf16_f32": late,
                       "loss_c_windows": _windows(curve["loss_c"]).round(4).tolist()}}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "training_equivalence_300_steps_dropout.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report))
    assert min(ref_acc) > 90.0 and min(acc) > 90.0, (ref_acc, acc)
    assert all(abs(a - b) <= tol.TRAIN_ACC_POINTS_DROPOUT for a, b in zip(acc, ref_acc)), (acc, ref_acc)
    assert dist["early"] <= tol.TRAIN_EARLY_REL_BF16, dist
    assert dist["late_loss_rel"] <= tol.TRAIN_LATE_REL_DROPOUT and dist["late_loss_a_rel"] <= tol.TRAIN_LATE_REL_DROPOUT, dist
    for k, (a, b) in late.items():      # under dropout both losses stay well above zero: compared relatively
        assert abs(a - b) <= tol.TRAIN_LATE_DROPOUT_REL * b + tol.TRAIN_LATE_DROPOUT_ABS, (k, a, b)
