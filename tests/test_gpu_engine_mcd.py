"""ens_DA MCD on TrainEngine (second video classifier + a second, gradient-reversed forward whose loss is the classifier
discrepancy on the target rows; main.py:447-448, 548-556): the reference's own trajectories - tests/golden/tiny_mcd,
tiny_mcd_noent (no attentive entropy, four steps), mid_dan_mcd (with the DAN discrepancy loss on top) - recorded from the unmodified
reference by tests/golden/make_golden.py: clipped gradients and parameters after every step, and the logged loss_s."""
import re

import pytest
import torch

from golden_util import Golden, case_config, step_schedule
from ta3n_amd import _lib
from ta3n_amd.engine import ALL_FLAGS, TrainEngine
from ta3n_amd.synthetic import synth_batch, synth_state

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["tiny_mcd", "tiny_mcd_noent", "mid_dan_mcd"])
def test_engine_with_mcd_follows_the_reference_trajectory(name):
    g = Golden(name)
    c = case_config(g)
    T, C = c["T"], c["C"]
    flags = ALL_FLAGS if c.get("add_loss_DA") != "none" else ALL_FLAGS & ~_lib.FLAG_ATTN_ENTROPY
    eng = TrainEngine(c["Bs"], c["Bt"], T, c["D"], c["fc_dim"], C, flags=flags, dropout_i=0.0, dropout_v=0.0, clip=c["clip"],
                      ens_DA="MCD", mu=c["mu"], dis_DA=c["dis_DA"], place_dis=c["place_dis"], alpha=c["alpha"])
    assert not eng.fused
    shapes = {n: s for n, _, s, _ in eng.plan.params}
    assert "fc_classifier_video_source_2.weight" in shapes
    eng.load_state(synth_state(shapes, seed=c["wseed"], scale=c["wscale"]))
    live = set(eng.live_names())
    assert live == set(str(k) for k in g.meta("live"))
    want_log = str(g.meta("log")).strip().splitlines()
    for s, st in enumerate(step_schedule(c)):
        xs, xt, ys, yt = synth_batch(C, T, c["D"], c["Bs"], c["Bt"], seed=st["xseed"])
        xs[st["n_src"]:] = 0; xt[st["n_tgt"]:] = 0
        eng.set_batch(xs.cuda(), xt.cuda(), ys.cuda())
        eng.train_step([0.75, 0.75, 0.5], 0.003, st["lr"], valid_source=st["n_src"], valid_target=st["n_tgt"])
        torch.cuda.synchronize()
        coef = eng.region("grad_norm")[1].item()
        grads = eng.param_views(eng.G)
        for k, v in eng.param_views().items():
            if k in live:
                g.check(f"step{s}/clipped_grad/{k}", grads[k].cpu() * coef, 2e-4, 5e-6, rms_atol=2e-4)
            g.check(f"step{s}/param/{k}", v.cpu(), 2e-4, 5e-6)
        m = re.search(r"loss_s\\s+(-?[0-9.]+)", want_log[s])
        if m:      # the reference's log line of this step (one step per "epoch": the running average is the step's value)
            assert abs(eng.loss_s.item() - float(m.group(1))) < 2e-4, (eng.loss_s.item(), want_log[s])
