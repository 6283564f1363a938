"""Two-stream wrapper (BASELINE configs[4]: two independent TA3N models on RGB- and Flow-shaped features, class logits summed):
the models stepped concurrently on two HIP streams give bit-for-bit the parameters of the same models stepped one after the
other, and of two stand-alone engines; the summed logits are the sum of the engines' logits."""
import pytest
import torch

from ta3n_amd.synthetic import synth_batch, synth_state

pytestmark = pytest.mark.gpu
SH = dict(Bs=12, Bt=8, T=5, D=(512, 256), F=64, C=12)


def _load(engs, seed0=5):
    for k, e in enumerate(engs):
        e.load_state(synth_state({n: s for n, _, s, _ in e.plan.params}, seed=seed0 + k))


def _batches(step):
    out = []
    for k, d in enumerate(SH["D"]):
        xs, xt, ys, _ = synth_batch(SH["C"], SH["T"], d, SH["Bs"], SH["Bt"], seed=100 * step + k)
        out.append((xs.cuda(), xt.cuda(), ys.cuda()))
    return out


@pytest.mark.parametrize("bf16", [False, True])
def test_concurrent_streams_equal_serial_and_standalone_engines(bf16):
    from ta3n_amd.engine import TrainEngine
    from ta3n_amd.two_stream import TwoStreamEngine
    kw = dict(dropout_i=0.5, dropout_v=0.5, bf16=bf16, bf16_store=bf16)
    two = {c: TwoStreamEngine(SH["Bs"], SH["Bt"], SH["T"], SH["D"], SH["F"], SH["C"], concurrent=c, **kw) for c in (True, False)}
    alone = [TrainEngine(SH["Bs"], SH["Bt"], SH["T"], d, SH["F"], SH["C"], **kw) for d in SH["D"]]
    for w in two.values():
        _load(w.streams)
    _load(alone)
    for step in range(3):
        b = _batches(step)
        for w in two.values():
            w.set_batch([x[0] for x in b], [x[1] for x in b], b[0][2])
            w.train_step([0.75, 0.75, 0.5], 0.003, 1e-2, seed=step)
        for e, (xs, xt, ys) in zip(alone, b):
            e.set_batch(xs, xt, b[0][2])             # (the wrapper feeds one label tensor to both streams)
            e.train_step_pipelined([0.75, 0.75, 0.5], 0.003, 1e-2, seed=step)
    for w in two.values():
        w.flush()
    for e in alone:
        e.flush()
    torch.cuda.synchronize()
    for k in range(2):
        assert torch.equal(two[True].streams[k].P, two[False].streams[k].P), (k, (two[True].streams[k].P - two[False].streams[k].P).abs().max())
        assert torch.equal(two[True].streams[k].P, alone[k].P), (k, (two[True].streams[k].P - alone[k].P).abs().max())
        assert bool(torch.isfinite(alone[k].P).all())
    want = alone[0].outputs()["out"] + alone[1].outputs()["out"]
    assert torch.equal(two[True].logits(), want)
    assert set(two[True].losses()) == set(alone[0].losses())


@pytest.mark.parametrize("concurrent", [True, False])
def test_one_library_call_for_both_models_equals_per_step_calls(concurrent):
    """ta3n_train_steps_multi (TwoStreamEngine.train_steps: K steps of both models from ONE call, each on its own HIP stream) against
    one train_step call per step and stream - bit for bit, starting both from a fresh engine (first step has no pending update) and
    continuing from a pending one."""
    from ta3n_amd.two_stream import TwoStreamEngine
    kw = dict(dropout_i=0.5, dropout_v=0.5, bf16=True, bf16_store=True)
    a = TwoStreamEngine(SH["Bs"], SH["Bt"], SH["T"], SH["D"], SH["F"], SH["C"], concurrent=concurrent, **kw)
    b = TwoStreamEngine(SH["Bs"], SH["Bt"], SH["T"], SH["D"], SH["F"], SH["C"], concurrent=True, **kw)
    _load(a.streams)
    _load(b.streams)
    bt = _batches(0)
    for w in (a, b):
        w.set_batch([x[0] for x in bt], [x[1] for x in bt], bt[0][2])
    sched = [([0.75, 0.75, 0.5], 0.003, 1e-2 / (1 + k)) for k in range(7)]
    a.train_steps(sched[:3])
    a.train_steps(sched[3:])
    for beta, gamma, lr in sched:
        b.train_step(beta, gamma, lr)
    a.flush()
    b.flush()
    torch.cuda.synchronize()
    for k in range(2):
        assert torch.equal(a.streams[k].P, b.streams[k].P), (k, (a.streams[k].P - b.streams[k].P).abs().max())
        assert torch.equal(a.streams[k].M, b.streams[k].M)
        assert a.streams[k].step_count == b.streams[k].step_count == 7
    assert torch.equal(a.logits(), b.logits())
