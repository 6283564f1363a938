"""The paper's "TemPooling + X" rows on the native loop: TrainEngine(aggregation='avgpool') with dis_DA DAN / JAN, ens_DA MCD and
use_bn AdaBN - the reference's own trajectories (tests/golden/tiny_avgpool_*: recorded from the unmodified reference by
tests/golden/make_golden.py): clipped gradients and parameters after every step."""
import pytest
import torch

from golden_util import Golden, case_config, step_schedule
from ta3n_amd.engine import TrainEngine, flags_from_options
from ta3n_amd.synthetic import synth_batch, synth_state

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["tiny_avgpool_dan_mcd", "tiny_avgpool_jan", "tiny_avgpool_adabn", "tiny_avgpool_mcd_noent"])
def test_tempooling_with_the_da_options_follows_the_reference_trajectory(name):
    g = Golden(name)
    c = case_config(g)
    T, C = c["T"], c["C"]
    assert c["agg"] == "avgpool"
    flags = flags_from_options(place_adv=c["place_adv"], add_loss_DA="none", use_attn="none")
    eng = TrainEngine(c["Bs"], c["Bt"], T, c["D"], c["fc_dim"], C, flags=flags, dropout_i=0.0, dropout_v=0.0, clip=c["clip"],
                      aggregation="avgpool", ens_DA=c["ens_DA"], mu=c["mu"], dis_DA=c["dis_DA"], place_dis=c["place_dis"], alpha=c["alpha"],
                      use_bn=c["use_bn"])
    shapes = {n: s for n, _, s, _ in eng.plan.params}
    eng.load_state(synth_state(shapes, seed=c["wseed"], scale=c["wscale"]))
    live = set(eng.live_names())
    assert live == set(str(k) for k in g.meta("live"))
    if c["use_bn"] != "none":      # the fixture's plain train-mode forward comes first and moves the BatchNorm buffers
        xs0, xt0, ys0, _ = synth_batch(C, T, c["D"], c["Bs"], c["Bt"], seed=c["xseed"])
        eng.set_batch(xs0.cuda(), xt0.cuda(), ys0.cuda())
        eng.set_hyper([0.75, 0.75, 0.5], 0.0, c["lr"], train=True)
        eng.forward()
    for s, st in enumerate(step_schedule(c)):
        xs, xt, ys, yt = synth_batch(C, T, c["D"], c["Bs"], c["Bt"], seed=st["xseed"])
        xs[st["n_src"]:] = 0; xt[st["n_tgt"]:] = 0
        eng.set_batch(xs.cuda(), xt.cuda(), ys.cuda())
        eng.train_step([0.75, 0.75, 0.5], 0.0, st["lr"], valid_source=st["n_src"], valid_target=st["n_tgt"])
        torch.cuda.synchronize()
        coef = eng.region("grad_norm")[1].item()
        grads = eng.param_views(eng.G)
        for k, v in eng.param_views().items():
            if k in live:
                g.check(f"step{s}/clipped_grad/{k}", grads[k].cpu() * coef, 2e-4, 5e-6, rms_atol=2e-4)
            g.check(f"step{s}/param/{k}", v.cpu(), 2e-4, 5e-6)
    if c["use_bn"] != "none":
        sd = eng.state_dict()
        for d in "ST":
            g.check(f"final/state/bn_shared_{d}.running_mean", sd[f"bn_shared_{d}.running_mean"].cpu(), 2e-4, 2e-5)
            g.check(f"final/state/bn_shared_{d}.running_var", sd[f"bn_shared_{d}.running_var"].cpu(), 2e-4, 2e-5)
