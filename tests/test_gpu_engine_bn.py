"""use_bn AdaBN / AutoDIAL on TrainEngine (the native train loop: HIP forward / loss / backward launch lists with the two BatchNorm
launches + fused clip / Nesterov update; the running statistics are engine buffers): the reference's own trajectories
(tests/golden/tiny_adabn, tiny_autodial, mid_adabn - recorded from the unmodified reference by tests/golden/make_golden.py):
parameters after every step, the BatchNorm buffers at the end, the eval-mode forward through them, checkpoint round trip."""
import pytest
import torch

import oracle.ta3n_oracle as orc
from golden_util import BN_CASES, Golden, case_config, step_schedule
from ta3n_amd.engine import TrainEngine
from ta3n_amd.synthetic import synth_batch, synth_state

pytestmark = pytest.mark.gpu


def _engine(c, fused=True, **kw):
    return TrainEngine(c["Bs"], c["Bt"], c["T"], c["D"], c["fc_dim"], c["C"], dropout_i=0.0, dropout_v=0.0, clip=c["clip"], use_bn=c["use_bn"],
                       fused=fused, **kw)


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "unfused"])
@pytest.mark.parametrize("name", BN_CASES)
def test_engine_with_domain_batchnorm_follows_the_reference_trajectory(name, fused):
    """fused (round 6, the default): ta3n_train_step with the two BatchNorm launches inside it (10 launches) and the fused gradient
    norm; unfused: the forward / loss / backward lists (17 launches) + a norm pass."""
    g = Golden(name)
    c = case_config(g)
    T, C = c["T"], c["C"]
    eng = _engine(c, fused)
    assert eng.fused == fused and eng.use_bn == c["use_bn"]
    shapes = {n: s for n, _, s, _ in eng.plan.params}
    assert "bn_shared_S.weight" in shapes and "bn_shared_T.bias" in shapes
    eng.load_state(synth_state(shapes, seed=c["wseed"], scale=c["wscale"]))
    live = set(eng.live_names())
    assert live == set(str(k) for k in g.meta("live"))
    xs0, xt0, ys0, _ = synth_batch(C, T, c["D"], c["Bs"], c["Bt"], seed=c["xseed"])
    # the fixture's plain train-mode forward comes first and moves the BatchNorm buffers like any train-mode pass
    eng.set_batch(xs0.cuda(), xt0.cuda(), ys0.cuda())
    eng.set_hyper([0.75, 0.75, 0.5], 0.003, c["lr"], train=True)
    eng.forward()
    o = eng.outputs()
    g.check("fwd/out_s", o["out"][:c["Bs"]].cpu(), 2e-4, 2e-4)
    g.check("fwd/feat_t_v", o["feat_v"][c["Bs"]:].cpu(), 2e-4, 2e-4)
    for s, st in enumerate(step_schedule(c)):
        xs, xt, ys, yt = synth_batch(C, T, c["D"], c["Bs"], c["Bt"], seed=st["xseed"])
        xs[st["n_src"]:] = 0; xt[st["n_tgt"]:] = 0            # the reference's dummy rows (main.py:359-364); BatchNorm sees them too
        eng.set_batch(xs.cuda(), xt.cuda(), ys.cuda())
        eng.train_step([0.75, 0.75, 0.5], 0.003, st["lr"], valid_source=st["n_src"], valid_target=st["n_tgt"])
        torch.cuda.synchronize()
        for k, v in eng.param_views().items():
            g.check(f"step{s}/param/{k}", v.cpu(), 2e-4, 5e-6)
    sd = eng.state_dict()
    for d in "ST":
        g.check(f"final/state/bn_shared_{d}.running_mean", sd[f"bn_shared_{d}.running_mean"].cpu(), 2e-4, 2e-5)
        g.check(f"final/state/bn_shared_{d}.running_var", sd[f"bn_shared_{d}.running_var"].cpu(), 2e-4, 2e-5)
        assert int(sd[f"bn_shared_{d}.num_batches_tracked"]) == int(g.z[f"final/state/bn_shared_{d}.num_batches_tracked#full"])
    # main.validate's eval-mode forward through the running statistics.  The fixture evaluated model(xs0, xs0): its target half are
    # the SOURCE videos normalised with the target statistics; eval-mode BatchNorm is row-independent, so the first min(Bs, Bt) rows do
    n = min(c["Bs"], c["Bt"])
    xt_eval = torch.zeros_like(xt0)
    xt_eval[:n] = xs0[:n]
    eng.set_batch(xs0.cuda(), xt_eval.cuda(), ys0.cuda())
    eng.set_hyper([0.0, 0.0, 0.0], 0.0, 0.0, train=False)
    eng.forward()
    torch.cuda.synchronize()
    ev = eng.outputs()
    for key, got in (("eval/out_t", ev["out"][c["Bs"]:c["Bs"] + n]), ("eval/feat_t_v", ev["feat_v"][c["Bs"]:c["Bs"] + n])):
        want = torch.from_numpy(g.z[key + "#full"]).float()[:n]
        assert torch.allclose(got.cpu().reshape(want.shape), want, rtol=3e-4, atol=3e-4), key
    assert eng.bn_batches == len(step_schedule(c)) + 1          # the eval pass does not move the buffers
    # a second engine restored from the state dict continues bit-identically
    eng2 = _engine(c, fused)
    eng2.load_state({k: v for k, v in sd.items()})
    eng2.M.copy_(eng.M)
    eng2.step_count = eng.step_count
    xs, xt, ys, yt = synth_batch(C, T, c["D"], c["Bs"], c["Bt"], seed=999)
    for e in (eng, eng2):
        e.set_batch(xs.cuda(), xt.cuda(), ys.cuda())
        e.train_step([0.75, 0.75, 0.5], 0.003, 1e-3, seed=5)
    torch.cuda.synchronize()
    assert torch.equal(eng.P, eng2.P) and torch.equal(eng.bn_running, eng2.bn_running) and eng.bn_batches == eng2.bn_batches


@pytest.mark.parametrize("name", ["tiny_adabn", "mid_adabn"])
def test_fused_batchnorm_step_pipelined_and_on_bf16_twins(name):
    """The fused BatchNorm step through the other ways a caller runs it: train_step_pipelined / train_steps (the update opens the next
    step; K steps in ONE library call - the BatchNorm launch moves the running statistics itself) must equal train_step bit for bit,
    running statistics included, and the bf16 arithmetic on twins (the BatchNorm launches keep the twins of F1 and gZ0) stays close to
    the fp32 step."""
    g = Golden(name)
    c = case_config(g)
    shapes = None
    runs = {}
    for how in ("plain", "pipelined", "steps", "bf16"):
        eng = _engine(c, True, **(dict(bf16=True, bf16_store=True) if how == "bf16" else {}))
        assert eng.fused and eng.can_batch_steps()      # (the BatchNorm launch tracks the running statistics on the device: K steps in one library call)
        shapes = {n: s for n, _, s, _ in eng.plan.params}
        eng.load_state(synth_state(shapes, seed=c["wseed"], scale=c["wscale"]))
        sched = []
        for s, st in enumerate(step_schedule(c)):
            sched.append(([0.75, 0.75, 0.5], 0.003, st["lr"]))
        xs, xt, ys, yt = synth_batch(c["C"], c["T"], c["D"], c["Bs"], c["Bt"], seed=step_schedule(c)[0]["xseed"])
        eng.set_batch(xs.cuda(), xt.cuda(), ys.cuda())
        if how in ("plain", "bf16"):
            for b_, g_, lr_ in sched:
                eng.train_step(b_, g_, lr_)
        elif how == "pipelined":
            for b_, g_, lr_ in sched:
                eng.train_step_pipelined(b_, g_, lr_)
            eng.flush()
        else:
            eng.train_steps(sched)
            eng.flush()
        torch.cuda.synchronize()
        runs[how] = (eng.P.clone(), eng.bn_running.clone(), eng.bn_batches)
    assert torch.equal(runs["plain"][0], runs["pipelined"][0]) and torch.equal(runs["plain"][1], runs["pipelined"][1])
    assert torch.equal(runs["plain"][0], runs["steps"][0]) and runs["plain"][2] == runs["steps"][2] == runs["pipelined"][2]
    d = (runs["bf16"][0] - runs["plain"][0]).norm() / (runs["plain"][0].norm() + 1e-30)
    assert torch.isfinite(runs["bf16"][0]).all() and d < 2e-2, d


@pytest.mark.parametrize("use_bn", ["AdaBN", "AutoDIAL"])
def test_tall_batches_stream_through_the_batchnorm_launches(use_bn):
    """More frame rows per domain than the BatchNorm launches keep in registers (1 024 row lanes x 5 rows = 5 120): their streaming
    loops re-read the column slab for the mean, the variance and the apply pass.  One train step at 5 500 / 5 200 rows against the
    oracle started from the same parameters: class logits, the BatchNorm affine gradients and every other gradient tensor."""
    Bs, Bt, T, D, Fc, Cn = 1100, 1040, 5, 64, 64, 7
    cfg = orc.Config(num_class=Cn, num_segments=T, feature_dim=D, fc_dim=Fc, dropout_i=0.0, dropout_v=0.0, use_bn=use_bn)
    params = synth_state(orc.param_shapes(cfg), seed=3, scale="trained")
    eng = TrainEngine(Bs, Bt, T, D, Fc, Cn, dropout_i=0.0, dropout_v=0.0, clip=20.0, use_bn=use_bn)
    assert eng.fused and Bs * T > 5120 and Bt * T > 5120
    eng.load_state(params)
    xs, xt, ys, yt = synth_batch(Cn, T, D, Bs, Bt, seed=17)
    state = orc.TrainState(params={k: v.detach().cpu().clone() for k, v in eng.param_views().items()}, lr=2e-3)
    state.momentum = {k: v.detach().cpu().clone() for k, v in eng.momentum_views().items()}
    eng.set_batch(xs.cuda(), xt.cuda(), ys.cuda())
    eng.train_step([0.75, 0.75, 0.5], 0.003, 2e-3)
    torch.cuda.synchronize()
    res = orc.train_step(state, xs, xt, ys, [0.75, 0.75, 0.5], 0.003, cfg, clip=20.0)
    want = torch.cat((res["src"]["out"], res["tgt"]["out"]), 0).detach()
    got = eng.outputs()["out"].cpu().reshape(want.shape)
    assert (got - want).abs().max().item() < 1e-3
    grads = eng.param_views(eng.G)
    top = max(w.double().norm().item() for w in res["grads"].values())
    for k, w in res["grads"].items():
        err, ref = (grads[k].cpu().double() - w.double()).norm().item(), w.double().norm().item()
        if ref < 1e-5 * top:      # the shared FC's bias in front of a BatchNorm: its gradient is zero in exact arithmetic, rounding noise in both
            assert err < 1e-5 * top, (k, err, ref)
        else:
            assert err < 5e-3 * ref, (k, err / ref)
    for k in ("bn_shared_S.weight", "bn_shared_S.bias", "bn_shared_T.weight", "bn_shared_T.bias"):
        assert k in res["grads"]
