"""Host-side schedules of the PRODUCT (ta3n_amd/engine.py) against values the reference itself produced, and the
statistical quality of the stateless dropout stream (ta3n_kernels.h: keep_mask, restated in tests/plan_interp.py; the GPU
tests check that the kernels use exactly that stream)."""
import numpy as np
import pytest

from golden_util import CASES, Golden, case_config, step_schedule
from plan_interp import keep_mask
from ta3n_amd.engine import beta_dann, dropout_seeds, lr_dann


@pytest.mark.parametrize("name", CASES)
def test_engine_lr_schedule_matches_the_learning_rate_the_reference_set(name):
    """make_golden records optimizer.param_groups[0]['lr'] after every reference step (adjust_learning_rate_dann, main.py:620-621,
    800-802) and the p it was computed from (main.py:350)."""
    g = Golden(name)
    c = case_config(g)
    for s, st in enumerate(step_schedule(c)):
        p = float(g.z[f"step{s}/p"][0])
        assert abs(lr_dann(c["lr"], p) - float(g.z[f"step{s}/lr_after"][0])) < 1e-12 * c["lr"] + 1e-15
        assert abs(p - st["p"]) < 1e-15


def test_engine_beta_schedule_is_main_py_351():
    for p in (0.0, 1e-3, 0.25, 0.5, 1.0):
        assert abs(beta_dann(p) - (2.0 / (1.0 + np.exp(-10 * p)) - 1)) < 1e-15
    assert beta_dann(0.0) == 0.0 and 0.9999 < beta_dann(1.0) < 1.0


def _corr(a, b):
    a = a - a.mean(); b = b - b.mean()
    return float((a * b).mean() / np.sqrt((a * a).mean() * (b * b).mean()))


def test_dropout_streams_are_independent():
    """keep-rate, independence of neighbouring elements, of the two streams of one step (dropout_i / dropout_v), of
    consecutive steps and of two ranks at the same step.  N = 2^20 draws: |correlation| of independent Bernoulli(1/2)
    masks is ~1e-3 (1 / sqrt N); bound 6e-3."""
    n = 1 << 20
    idx = np.arange(n)
    si0, sv0 = dropout_seeds(0, 0)
    si1, sv1 = dropout_seeds(1, 0)
    ri0, rv0 = dropout_seeds(0, 1)
    assert len({si0, sv0, si1, sv1, ri0, rv0}) == 6
    m = {k: keep_mask(s, idx, 0.5) for k, s in dict(i0=si0, v0=sv0, i1=si1, r0=ri0).items()}
    for k, v in m.items():
        assert abs(v.mean() - 0.5) < 3e-3, (k, v.mean())
    for lag in (1, 2, 3, 64, 511, 512, 513, 2048):            # neighbours along a row and across rows (F = 512)
        assert abs(_corr(m["i0"][:-lag], m["i0"][lag:])) < 6e-3, lag
    assert abs(_corr(m["i0"], m["v0"])) < 6e-3               # the two streams of a step
    assert abs(_corr(m["i0"], m["i1"])) < 6e-3               # consecutive steps
    assert abs(_corr(m["i0"], m["r0"])) < 6e-3               # two ranks, same step, same LOCAL indices
    for p in (0.1, 0.8):
        assert abs(keep_mask(si0, idx, p).mean() - (1 - p)) < 3e-3
    # runs: the number of sign changes of an independent sequence is n/2 +- a few sqrt(n)/2
    changes = np.count_nonzero(np.diff(m["i0"]))
    assert abs(changes - n / 2) < 4 * np.sqrt(n) / 2


@pytest.mark.parametrize("rank,world,step0", [(0, 1, 0), (3, 8, 17), (1, 2, 2 ** 31 - 3), (5, 8, 3_000_000_011)])
def test_schedule_array_of_a_multi_step_call_equals_the_per_step_scalars(rank, world, step0):
    """TrainEngine.hyper_array (the ta3n_hyper array one ta3n_train_steps call carries, filled column-wise) against hyper_for entry by
    entry (what set_hyper uploads for a single step): the same bytes - beta / gamma / lr rounded to fp32 the same way, the dropout
    seeds of dropout_seeds(step, rank) also where the 64-bit products wrap.  The engine's device state is not involved: the two
    methods run on a stand-in that carries the attributes they read."""
    import ctypes as C
    import types

    from ta3n_amd import _lib
    from ta3n_amd.engine import TrainEngine

    eng = types.SimpleNamespace(Bs=128, Bt=74, world=world, rank=rank, momentum=0.9, weight_decay=1e-4, clip=20.0, dropout_i=0.5,
                                dropout_v=0.3, step_count=5, T=9, _hyper=_lib.Hyper(), _global_source=0, _global_target=0)
    for name in ("set_hyper", "hyper_for", "hyper_array"):
        setattr(eng, name, types.MethodType(getattr(TrainEngine, name), eng))
    rng = np.random.default_rng(step0 % 1000)
    for m in (0, 1, 2, 37):
        entries = [([beta_dann(p), 0.75 * beta_dann(p), float(rng.random())], float(rng.random()) * 0.01, lr_dann(3e-2, p))
                   for p in rng.random(m)]
        hy = eng.hyper_array(entries, step0)
        assert len(hy) == m
        for k, (beta, gamma, lr) in enumerate(entries):
            want = eng.hyper_for(beta, gamma, lr, step=step0 + k)
            assert bytes(hy[k]) == bytes(want), (m, k)
            assert (hy[k].seed_i, hy[k].seed_v) == dropout_seeds(step0 + k, rank)
        assert C.sizeof(hy) == m * C.sizeof(_lib.Hyper)
