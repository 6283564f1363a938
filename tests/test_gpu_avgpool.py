"""BASELINE configs[0] on the GPU: TemPooling (frame_aggregation 'avgpool'), source-only - hmdb_ucf_small's shape.
The HIP path (TA3N_AGG_AVGPOOL: F1 GEMM, pool_cls kernel, {dWsh, dWcv} GEMM, SGD) through the C ABI against the fixtures the
reference itself produced (tests/golden/make_golden.py: models.VideoModel(..., 'avgpool', ...) + main.train with
use_target none).  fp32 tolerance as in test_gpu_parity.py: logits within 1e-3."""
import pytest
import torch

from golden_util import AVG_CASES, Golden, case_config, step_schedule
from ta3n_amd.synthetic import synth_batch, synth_state

pytestmark = pytest.mark.gpu

LOGIT_ATOL = 1e-3
RTOL, ATOL = 2e-4, 5e-5


def _engine(c, **kw):
    from ta3n_amd.engine import TrainEngine
    return TrainEngine(c["Bs"], c["Bt"], c["T"], c["D"], c["fc_dim"], c["C"], dropout_i=0.0, dropout_v=0.0, clip=c["clip"],
                       aggregation="avgpool", **kw)


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("name", AVG_CASES)
def test_avgpool_train_steps_match_reference_golden(name, fused):
    g = Golden(name)
    c = case_config(g)
    eng = _engine(c, fused=fused)
    assert eng.plan.has_fused_step
    shapes = {n: s for n, _, s, _ in eng.plan.params}
    eng.load_state(synth_state(shapes, seed=c["wseed"], scale=c["wscale"]))
    live = set(eng.live_names())
    assert live == set(str(k) for k in g.meta("live"))
    B, Bs, T = c["Bs"] + c["Bt"], c["Bs"], c["T"]
    for s, st in enumerate(step_schedule(c)):
        xs, xt, ys, yt = synth_batch(c["C"], T, c["D"], c["Bs"], c["Bt"], seed=st["xseed"])
        xs[st["n_src"]:] = 0; xt[st["n_tgt"]:] = 0
        eng.set_batch(xs.cuda(), xt.cuda(), ys.cuda())
        eng.set_hyper([0.0, 0.0, 0.0], 0.0, st["lr"], train=True, valid_source=st["n_src"], valid_target=st["n_tgt"])
        if fused:
            eng.fused_step()
        else:
            eng.forward()
        if s == 0:
            o = {k: v.detach().cpu() for k, v in eng.outputs().items()}
            for dom, sl in (("s", slice(0, Bs)), ("t", slice(Bs, B))):
                g.check(f"fwd/out_{dom}", o["out"][sl], 0, LOGIT_ATOL, "class logits")
                g.check(f"fwd/attn_{dom}", o["attn"][sl], RTOL, ATOL)
                g.check(f"fwd/feat_{dom}_v", o["feat_v"][sl], RTOL, ATOL)
                g.check(f"fwd/feat_{dom}_f1", o["feat_f1"][sl], RTOL, ATOL)
        if not fused:
            eng.loss()
            eng.backward()
        raw = {k: v.clone() for k, v in eng.param_views(eng.G).items()}
        if fused:
            eng.sgd_step_fused()
        else:
            eng.sgd_step()
        torch.cuda.synchronize()
        coef = eng.region("grad_norm")[1].item()
        new = eng.param_views()
        for k in new:
            if k in live:
                g.check(f"step{s}/clipped_grad/{k}", raw[k].cpu() * coef, 1e-3, 2e-5, rms_atol=1e-2 if s == 0 else 0.15)
            g.check(f"step{s}/param/{k}", new[k].cpu(), RTOL, ATOL)
    # the logging scalar: the reference's "Loss" column of the last step (CE on the source rows)
    last = [ln for ln in str(g.meta("log")).strip().splitlines() if "Loss" in ln][-1]
    ref_loss = float(last.split("loss_c")[1].split()[0])
    assert abs(eng.losses()["loss_c"] - ref_loss) < 2e-3 and abs(eng.losses()["loss"] - ref_loss) < 2e-3


def test_avgpool_bf16_and_dropout_run():
    """bf16 arithmetic (operands rounded to bf16, twins where the launch allows) and dropout on: finite, close to fp32."""
    g = Golden("config1_avgpool")
    c = case_config(g)
    res = []
    for bf16 in (False, True):
        from ta3n_amd.engine import TrainEngine
        eng = TrainEngine(c["Bs"], c["Bt"], c["T"], c["D"], c["fc_dim"], c["C"], dropout_i=0.5, dropout_v=0.5, clip=c["clip"],
                          aggregation="avgpool", bf16=bf16, bf16_store=bf16)
        shapes = {n: s for n, _, s, _ in eng.plan.params}
        eng.load_state(synth_state(shapes, seed=c["wseed"], scale=c["wscale"]))
        xs, xt, ys, yt = synth_batch(c["C"], c["T"], c["D"], c["Bs"], c["Bt"], seed=3)
        eng.set_batch(xs.cuda(), xt.cuda(), ys.cuda())
        for i in range(3):
            eng.train_step_pipelined([0.0, 0.0, 0.0], 0.0, 1e-2, seed=i)
        eng.flush()
        torch.cuda.synchronize()
        res.append((eng.P.clone(), eng.outputs()["out"].clone(), eng.losses()["loss"]))
    (p32, o32, l32), (p16, o16, l16) = res
    assert torch.isfinite(p16).all() and l32 > 0
    assert (o32 - o16).abs().max().item() <= 0.1 * o32.pow(2).mean().sqrt().item()
    assert abs(l32 - l16) <= 0.05 * abs(l32)
