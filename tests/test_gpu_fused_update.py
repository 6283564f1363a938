"""The optimiser inside the gradient launches (ta3n_train_steps_fused_update; TrainEngine.train_steps(fused_update=True)): every
gradient tile applies the Nesterov / weight-decay step to its own block of parameters, two parameter buffers alternate, one short
launch per step checks the clip norm (and corrects the rare step that clips).
 * no step clips  -> BIT-identical parameters, momentum, gradients, losses, twins to the update as launches of its own;
 * steps that clip -> equal within fp32 rounding of the linear correction, and equal to the reference's goldens (tiny_clip);
 * odd / even step counts (the result always lands in the caller's buffer), repeated calls, device-side batch feeds."""
import pytest
import torch

from golden_util import Golden, case_config, step_schedule
from ta3n_amd.engine import TrainEngine
from ta3n_amd.synthetic import synth_batch, synth_state

pytestmark = pytest.mark.gpu_ab      # measured-and-rejected variant / opt-in transport: `pytest -m gpu_ab` on the experiments build (tests/conftest.py)

ARITH = {"f32": {}, "bf16": dict(bf16=True, bf16_store=True), "bf16_cvt": dict(bf16=True), "f32x3": dict(f32_split=True)}


def _run(shape, arith, fused_update, n_steps, clip, calls=1, dropout=0.5, lr=2e-3):
    Bs, Bt, T, D, F, C = shape
    eng = TrainEngine(Bs, Bt, T, D, F, C, dropout_i=dropout, dropout_v=dropout, clip=clip, **ARITH[arith])
    eng.load_state(synth_state({n: s for n, _, s, _ in eng.plan.params}, seed=7))
    xs, xt, ys, yt = synth_batch(C, T, D, Bs, Bt, seed=5)
    eng.set_batch(xs.cuda(), xt.cuda(), ys.cuda())
    sched = [([0.75, 0.75, 0.5], 0.003, lr * (1 + 0.1 * i)) for i in range(n_steps)]
    per = (n_steps + calls - 1) // calls
    coefs = []
    for c0 in range(0, n_steps, per):
        eng.train_steps(sched[c0:c0 + per], fused_update=fused_update)
        eng.flush()
        coefs.append(eng.region("grad_norm")[1].item())
    torch.cuda.synchronize()
    assert eng.step_count == n_steps
    out = dict(P=eng.P.clone(), M=eng.M.clone(), G=eng.G[: eng.plan.live_floats].clone(), losses=eng.region("losses")[:6].clone())
    if eng.bf16_store:
        off, n = eng.plan.region("p16")
        out["p16"] = eng.ws[off:off + n].view(torch.int16)[: eng.plan.param_floats].clone()
    return out, coefs


@pytest.mark.parametrize("shape,arith,n_steps,calls",
                         [(sh, a, n, c) for sh in [(6, 4, 5, 512, 64, 12), (33, 37, 9, 192, 64, 30), (128, 74, 5, 2048, 512, 12)] for a in sorted(ARITH)
                          for n, c in [(1, 1), (4, 1), (5, 1), (7, 3)] if sh[0] != 128 or (n, c) in ((5, 1), (7, 3))])      # (headline: two schedules)
def test_fused_update_is_bit_identical_when_no_step_clips(shape, arith, n_steps, calls):
    a, ca = _run(shape, arith, False, n_steps, clip=1e6, calls=calls)
    b, cb = _run(shape, arith, True, n_steps, clip=1e6, calls=calls)
    assert all(c == 1.0 for c in ca + cb)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert torch.isfinite(a["P"]).all()
    if "p16" in b:      # the twins of the result: RNE_bf16(parameters), in the region every other entry point reads
        live = b["P"].numel()
        want = b["P"].to(torch.bfloat16).view(torch.int16)
        assert torch.equal(b["p16"][:live], want)


@pytest.mark.parametrize("arith", ["f32", "bf16"])
def test_fused_update_corrects_steps_that_clip(arith):
    """clip far below the gradient norm: EVERY step clips; the speculative update + linear correction must equal the direct
    computation up to fp32 rounding of the correction."""
    shape = (6, 4, 5, 512, 64, 12)
    a, ca = _run(shape, arith, False, 5, clip=0.05, dropout=0.0)
    b, cb = _run(shape, arith, True, 5, clip=0.05, dropout=0.0)
    assert ca[-1] < 1.0 and abs(ca[-1] - cb[-1]) < 1e-5 * ca[-1]
    assert torch.allclose(a["P"], b["P"], rtol=2e-5, atol=1e-7), (a["P"] - b["P"]).abs().max()
    assert torch.allclose(a["M"], b["M"], rtol=2e-4, atol=1e-6), (a["M"] - b["M"]).abs().max()


@pytest.mark.parametrize("name", ["tiny_T5", "tiny_clip", "tiny_T9"])
def test_fused_update_matches_reference_golden(name):
    """The reference's recorded parameters after each step (tiny_clip: the clip branch is active), one fused-update call per step."""
    from ta3n_amd import tolerances as tol
    g = Golden(name)
    c = case_config(g)
    eng = TrainEngine(c["Bs"], c["Bt"], c["T"], c["D"], c["fc_dim"], c["C"], dropout_i=0.0, dropout_v=0.0, clip=c["clip"])
    eng.load_state(synth_state({n: s for n, _, s, _ in eng.plan.params}, seed=c["wseed"], scale=c["wscale"]))
    clipped = False
    for s, st in enumerate(step_schedule(c)):
        xs, xt, ys, yt = synth_batch(c["C"], c["T"], c["D"], c["Bs"], c["Bt"], seed=st["xseed"])
        xs[st["n_src"]:] = 0; xt[st["n_tgt"]:] = 0
        eng.set_batch(xs.cuda(), xt.cuda(), ys.cuda())
        h = dict(valid_source=st["n_src"], valid_target=st["n_tgt"])
        keep = eng.hyper_for
        eng.hyper_for = lambda b, ga, lr, step=None, _k=keep, _h=h: _k(b, ga, lr, step=step, **_h)      # (ragged last batches: valid rows travel in the scalars)
        eng.train_steps([([0.75, 0.75, 0.5], 0.003, st["lr"])], fused_update=True)
        eng.hyper_for = keep
        torch.cuda.synchronize()
        clipped = clipped or eng.region("grad_norm")[1].item() < 1.0
        for k, v in eng.param_views().items():
            g.check(f"step{s}/param/{k}", v.cpu(), tol.F32_RTOL, tol.F32_ATOL)
    if name == "tiny_clip":
        assert clipped              # the correction path ran and still landed on the reference's parameters


def test_fused_update_with_device_side_batch_feeds(tmp_path):
    from test_gpu_train_steps import _make_store
    Bs, Bt, T, D, F, C = 6, 4, 5, 512, 64, 7
    src, tgt = _make_store(tmp_path, "s", 14, D, "f32"), _make_store(tmp_path, "t", 9, D, "f32")
    gen = torch.Generator().manual_seed(3)
    n = 5
    ids_s = torch.stack([torch.randperm(14, generator=gen)[:Bs] for _ in range(n)]).to(torch.int32).cuda()
    ids_t = torch.stack([torch.randperm(9, generator=gen)[:Bt] for _ in range(n)]).to(torch.int32).cuda()
    sched = [([0.75, 0.75, 0.5], 0.003, 2e-3) for _ in range(n)]
    res = []
    for fu in (False, True):
        eng = TrainEngine(Bs, Bt, T, D, F, C, dropout_i=0.5, dropout_v=0.5, clip=1e6, bf16=True, bf16_store=True)
        eng.load_state(synth_state({n_: s for n_, _, s, _ in eng.plan.params}, seed=7))
        eng.train_steps(sched, feeds=((src, ids_s), (tgt, ids_t)), fused_update=fu)
        eng.flush()
        torch.cuda.synchronize()
        res.append((eng.P.clone(), eng.M.clone(), eng.region("losses")[:6].clone()))
    assert all(torch.equal(x, y) for x, y in zip(*res))
