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


@pytest.mark.parametrize("entropy", [True, False], ids=["attentive_entropy", "no_entropy"])
@pytest.mark.parametrize("ns,nt", [(24, 17), (9, 0), (0, 12), (128, 74)])      # (the last: BASELINE configs[1]'s full shape)
def test_native_mcd_assembly_matches_the_torch_form(entropy, ns, nt, monkeypatch):
    """ta3n_mcd_source_loss / ta3n_mcd_second_loss (the library's kernels) against the torch assembly they replace (TA3N_NATIVE_MCD=0): the
    second classifier's cross-entropy, loss_s, the moved entropy term and - through three whole steps with dropout, i.e. different masks in
    the two passes - every parameter; ragged valid rows, an empty domain."""
    Bs, Bt, T, D, F, C = (128, 74, 5, 2048, 512, 12) if (ns, nt) == (128, 74) else (24, 20, 3, 64, 64, 7)
    flags = ALL_FLAGS if entropy else ALL_FLAGS & ~_lib.FLAG_ATTN_ENTROPY
    runs = {}
    for native in ("1", "0"):
        monkeypatch.setenv("TA3N_NATIVE_MCD", native)
        eng = TrainEngine(Bs, Bt, T, D, F, C, flags=flags, dropout_i=0.5, dropout_v=0.5, clip=20.0, ens_DA="MCD", mu=0.4)
        assert eng._mcd_native == (native == "1")
        eng.load_state(synth_state({n: s for n, _, s, _ in eng.plan.params}, seed=5, scale="trained"))
        logs = []
        for s in range(3):
            xs, xt, ys, yt = synth_batch(C, T, D, Bs, Bt, seed=30 + s)
            xs[ns:] = 0; xt[nt:] = 0
            eng.set_batch(xs.cuda(), xt.cuda(), ys.cuda())
            eng.train_step([0.75, 0.75, 0.5], 0.003, 1e-2, valid_source=ns, valid_target=nt, seed=s)
            torch.cuda.synchronize()
            sh = eng.loss_e_shift
            logs.append((float(eng.loss_c2), float(eng.loss_s), None if sh is None else (float(sh[0]), float(sh[1]))))
        runs[native] = (eng.P.clone().cpu(), logs)
    (p1, l1), (p0, l0) = runs["1"], runs["0"]
    for a, b in zip(l1, l0):
        assert abs(a[0] - b[0]) <= 1e-5 * max(1.0, abs(b[0])) and abs(a[1] - b[1]) <= 1e-6 + 1e-5 * abs(b[1]), (a, b)
        assert (a[2] is None) == (b[2] is None), (a, b)
        if a[2] is not None:
            assert abs(a[2][0] - b[2][0]) <= 1e-6 + 1e-4 * abs(b[2][0]) and abs(a[2][1] - b[2][1]) <= 1e-5 + 1e-4 * abs(b[2][1]), (a, b)
    assert (p1 - p0).norm().item() <= 2e-6 * p0.norm().item(), (p1 - p0).norm().item() / p0.norm().item()
