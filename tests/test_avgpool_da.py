"""TemPooling + RevGrad (first item): frame_aggregation 'avgpool' with the video- and frame-level
adversarial branches, against fixtures produced by the reference itself (tests/golden/make_golden.py: adv_DA RevGrad,
use_target uSv, use_attn none, the fixture's place_adv) - oracle on the CPU, the launch plan executed with numpy on the
CPU, the HIP path on the GPU.  With place_adv[0] = 'Y' the reference counts the video-level loss twice (without relation
features its relation slot holds the video logits again, models.py:707-708): reproduced, and covered by tiny_avgpool_da3."""
import numpy as np
import pytest
import torch

from golden_util import AVG_DA_CASES, AVG_DA_EXTRA_CASES, Golden, case_config, step_schedule
from oracle import ta3n_oracle as orc
from plan_interp import Interp
from ta3n_amd import _lib
from ta3n_amd.engine import flags_from_options
from ta3n_amd.synthetic import synth_batch, synth_state
from test_plan_cpu import make_hyper

RTOL, ATOL = 2e-5, 2e-6
BETA = [0.75, 0.75, 0.5]


def _setup(name):
    g = Golden(name)
    c = case_config(g)
    assert c["agg"] == "avgpool" and c["place_adv"] is not None
    cfg = orc.Config(num_class=c["C"], num_segments=c["T"], feature_dim=c["D"], fc_dim=c["fc_dim"], dropout_i=0.0, dropout_v=0.0,
                     place_adv=c["place_adv"], add_loss_DA="none", use_attn="none", frame_aggregation="avgpool",
                     dis_DA=c["dis_DA"], place_dis=c["place_dis"], ens_DA=c["ens_DA"], use_bn=c["use_bn"])
    params = synth_state(orc.param_shapes(cfg), seed=c["wseed"], scale=c["wscale"])
    flags = flags_from_options(c["place_adv"], "none", "none", "RevGrad", "uSv")
    return g, c, cfg, params, flags


@pytest.mark.parametrize("name", AVG_DA_CASES + AVG_DA_EXTRA_CASES)
def test_oracle_matches_reference(name):
    g, c, cfg, params, _ = _setup(name)
    xs, xt, ys, yt = synth_batch(c["C"], c["T"], c["D"], c["Bs"], c["Bt"], seed=c["xseed"])
    with torch.no_grad():
        s = orc.forward_domain(params, xs, BETA, cfg, domain="S")
        t = orc.forward_domain(params, xt, BETA, cfg, domain="T")
    for dom, o in (("s", s), ("t", t)):
        g.check(f"fwd/out_{dom}", o["out"], RTOL, ATOL)
        for i, nm in enumerate(("rel", "vid", "frm")):
            g.check(f"fwd/pd_{dom}_{nm}", o["pred_domain"][i], RTOL, ATOL)
        for i, nm in enumerate(("y", "v", "f1")):
            g.check(f"fwd/feat_{dom}_{nm}", o["feat"][i], RTOL, ATOL)
    state = orc.TrainState(params=params, lr=c["lr"])
    live = set(str(k) for k in g.meta("live"))
    for si, st in enumerate(step_schedule(c)):
        xs, xt, ys, yt = synth_batch(c["C"], c["T"], c["D"], c["Bs"], c["Bt"], seed=st["xseed"])
        xs[st["n_src"]:] = 0; xt[st["n_tgt"]:] = 0
        state.lr = st["lr"]
        res = orc.train_step(state, xs, xt, ys, BETA, 0.0, cfg, clip=c["clip"], n_src=st["n_src"], n_tgt=st["n_tgt"],
                             alpha=c["alpha"], mu=c["mu"])
        assert set(res["clipped"]) == live
        for k in params:
            if k in live:
                g.check(f"step{si}/clipped_grad/{k}", res["clipped"][k], 5e-5, 5e-6)
            g.check(f"step{si}/param/{k}", state.params[k], 5e-5, 5e-6)


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("name", [n for n in AVG_DA_CASES if n.startswith("tiny")])
def test_plan_reproduces_reference_on_cpu(name, fused):
    g, c, _, _, flags = _setup(name)
    T = c["T"]
    plan = _lib.Plan(c["Bs"], c["Bt"], T, c["D"], c["fc_dim"], c["C"], flags, aggregation=_lib.AGG_AVGPOOL)
    assert plan.has_fused_step
    it = Interp(plan)
    shapes = {n: s for n, _, s, _ in plan.params}
    it.set_params(synth_state(shapes, seed=c["wseed"], scale=c["wscale"]))
    live = {n for n, _, _, lv in plan.params if lv}
    assert live == set(str(k) for k in g.meta("live"))
    for s, st in enumerate(step_schedule(c)):
        xs, xt, ys, yt = synth_batch(c["C"], T, c["D"], c["Bs"], c["Bt"], seed=st["xseed"])
        xs[st["n_src"]:] = 0; xt[st["n_tgt"]:] = 0
        it.X = torch.cat((xs, xt), 0).double().numpy().reshape(-1)
        it.labels[:c["Bs"]] = ys.numpy()
        it.hy = make_hyper(c, st, T, st["lr"])
        it.hy["gamma"] = 0.0
        it.G[:] = 0
        if fused:
            it.run_group(4)
        else:
            it.run_group(0); it.run_group(1); it.run_group(2)
        if s == 0:
            B, Bs = c["Bs"] + c["Bt"], c["Bs"]
            geo = it.g
            outs = dict(out=it.r(geo.o_Y, (B, c["C"])), vid=it.r(geo.o_Pv, (B, 2)), frm=it.r(geo.o_Pf, (B, T, 2)), v=it.r(geo.o_V, (B, geo.F)))
            for dom, sl in (("s", slice(0, Bs)), ("t", slice(Bs, B))):
                g.check(f"fwd/out_{dom}", outs["out"][sl], 5e-5, 2e-5)
                g.check(f"fwd/feat_{dom}_v", outs["v"][sl], 5e-5, 2e-5)
                if c["place_adv"][1] == "Y" or c["place_adv"][0] == "Y":
                    g.check(f"fwd/pd_{dom}_vid", outs["vid"][sl], 5e-5, 2e-5)
                    g.check(f"fwd/pd_{dom}_rel", outs["vid"][sl], 5e-5, 2e-5)     # the relation slot IS the video logits
                if c["place_adv"][2] == "Y":
                    g.check(f"fwd/pd_{dom}_frm", outs["frm"][sl], 5e-5, 2e-5)
        raw = it.get_params(it.G)
        it.run_group(3, fused_norm=fused)
        coef = it.ws[it.g.o_grad_norm + 1]
        new = it.get_params()
        for k in shapes:
            if k in live:
                g.check(f"step{s}/clipped_grad/{k}", raw[k] * coef, 1e-4, 2e-5)
            g.check(f"step{s}/param/{k}", new[k], 1e-4, 2e-5)


def test_fused_tempooling_plan_with_domain_batchnorm_on_cpu():
    """use_bn on TemPooling through the FUSED list (round 6: TrainEngine runs use_bn fused): the reference's tiny_avgpool_adabn trajectory,
    and the sumsq slots hold the whole gradient norm - the BatchNorm gradients' share in the LAST slots, no tile's slot overwritten."""
    g, c, _, _, flags = _setup("tiny_avgpool_adabn")
    T = c["T"]
    plan = _lib.Plan(c["Bs"], c["Bt"], T, c["D"], c["fc_dim"], c["C"], flags | _lib.FLAG_BN_SHARED, aggregation=_lib.AGG_AVGPOOL)
    assert plan.has_fused_step
    it = Interp(plan)
    shapes = {n: s for n, _, s, _ in plan.params}
    it.set_params(synth_state(shapes, seed=c["wseed"], scale=c["wscale"]))
    live = {n for n, _, _, lv in plan.params if lv}
    assert live == set(str(k) for k in g.meta("live"))
    for s, st in enumerate(step_schedule(c)):
        xs, xt, ys, yt = synth_batch(c["C"], T, c["D"], c["Bs"], c["Bt"], seed=st["xseed"])
        xs[st["n_src"]:] = 0; xt[st["n_tgt"]:] = 0
        it.X = torch.cat((xs, xt), 0).double().numpy().reshape(-1)
        it.labels[:c["Bs"]] = ys.numpy()
        it.hy = make_hyper(c, st, T, st["lr"])
        it.hy["gamma"] = 0.0
        it.G[:] = 0
        it.run_group(4)
        total = np.sqrt(sum(float((v.astype(np.float64) ** 2).sum()) for k, v in it.get_params(it.G).items() if k in live))
        slots = np.sqrt(it.ws[it.g.o_sumsq:it.g.o_sumsq + it.g.n_sumsq].sum())
        assert abs(slots - total) <= 1e-6 * total, (slots, total)
        it.run_group(3, fused_norm=True)
        new = it.get_params()
        for k in shapes:
            g.check(f"step{s}/param/{k}", new[k], 1e-4, 2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("name", AVG_DA_CASES)
def test_hip_path_matches_reference_golden(name, fused):
    from ta3n_amd.engine import TrainEngine
    g, c, _, _, flags = _setup(name)
    eng = TrainEngine(c["Bs"], c["Bt"], c["T"], c["D"], c["fc_dim"], c["C"], flags=flags, dropout_i=0.0, dropout_v=0.0, clip=c["clip"],
                      aggregation="avgpool", fused=fused)
    shapes = {n: s for n, _, s, _ in eng.plan.params}
    eng.load_state(synth_state(shapes, seed=c["wseed"], scale=c["wscale"]))
    live = set(eng.live_names())
    assert live == set(str(k) for k in g.meta("live"))
    B, Bs, T = c["Bs"] + c["Bt"], c["Bs"], c["T"]
    for s, st in enumerate(step_schedule(c)):
        xs, xt, ys, yt = synth_batch(c["C"], T, c["D"], c["Bs"], c["Bt"], seed=st["xseed"])
        xs[st["n_src"]:] = 0; xt[st["n_tgt"]:] = 0
        eng.set_batch(xs.cuda(), xt.cuda(), ys.cuda())
        eng.set_hyper(BETA, 0.0, st["lr"], train=True, valid_source=st["n_src"], valid_target=st["n_tgt"])
        if fused:
            eng.fused_step()
        else:
            eng.forward(); eng.loss(); eng.backward()
        if s == 0:
            o = {k: v.detach().cpu() for k, v in eng.outputs().items()}
            for dom, sl in (("s", slice(0, Bs)), ("t", slice(Bs, B))):
                g.check(f"fwd/out_{dom}", o["out"][sl], 0, 1e-3, "class logits")          # the north-star bound
                g.check(f"fwd/feat_{dom}_v", o["feat_v"][sl], 2e-4, 5e-5)
                if c["place_adv"][1] == "Y" or c["place_adv"][0] == "Y":
                    g.check(f"fwd/pd_{dom}_vid", o["pred_vid"][sl], 0, 1e-3, "video domain logits")
                if c["place_adv"][2] == "Y":
                    g.check(f"fwd/pd_{dom}_frm", o["pred_frm"][sl], 0, 1e-3, "frame domain logits")
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
            g.check(f"step{s}/param/{k}", new[k].cpu(), 2e-4, 5e-5)
    last = [ln for ln in str(g.meta("log")).strip().splitlines() if "Loss" in ln][-1]       # the reference's own log line
    ref_loss = float(last.split("Loss")[1].split()[0])
    ref_a = float(last.split("loss_a")[1].split()[0])
    L = eng.losses()
    assert abs(L["loss"] - ref_loss) < 3e-3 and abs(L["loss_adv_vid"] + L["loss_adv_frm"] + L["loss_adv_rel"] - ref_a) < 3e-3


@pytest.mark.gpu
@pytest.mark.parametrize("place_adv", [("N", "Y", "Y"), ("Y", "Y", "Y"), ("N", "N", "N")])
def test_module_path_avgpool_matches_oracle(place_adv):
    """VideoModel(frame_aggregation='avgpool'): forward 10-tuple and, with the loss assembled by the CALLER as main.train does
    (main.py:439-451, 508-538), every parameter gradient - against the oracle; unused discriminators keep grad None."""
    import torch.nn.functional as Fn
    from ta3n_amd.models import VideoModel
    C_, T, D, Fc, Bs, Bt = 7, 4, 512, 64, 6, 5
    cfg = orc.Config(num_class=C_, num_segments=T, feature_dim=D, fc_dim=Fc, dropout_i=0.0, dropout_v=0.0, place_adv=place_adv,
                     add_loss_DA="none", use_attn="none", frame_aggregation="avgpool")
    params = synth_state(orc.param_shapes(cfg), seed=21)
    xs, xt, ys, yt = synth_batch(C_, T, D, Bs, Bt, seed=22)
    res = orc.train_step(orc.TrainState(params={k: v.clone() for k, v in params.items()}), xs, xt, ys, BETA, 0.0, cfg, clip=None)
    m = VideoModel(C_, "video", "avgpool", "RGB", train_segments=T, val_segments=T, base_model="resnet18", fc_dim=Fc, dropout_i=0.0,
                   dropout_v=0.0, use_attn="none", verbose=False)
    sd = m.state_dict(); sd.update(params); m.load_state_dict(sd)
    m = m.cuda(); m.train()
    out = m(xs, xt, BETA, 0, True, False)
    attn_s, out_s, out_s2, pd_s, feat_s, attn_t, out_t, out_t2, pd_t, feat_t = out
    assert torch.allclose(out_s.cpu(), res["src"]["out"], atol=1e-3) and torch.allclose(out_t.cpu(), res["tgt"]["out"], atol=1e-3)
    assert torch.allclose(pd_s[1].cpu(), res["src"]["pred_domain"][1], atol=1e-3) and torch.equal(pd_s[0], pd_s[1])
    assert torch.allclose(pd_t[2].cpu(), res["tgt"]["pred_domain"][2], atol=1e-3)
    assert torch.allclose(attn_s.cpu(), res["src"]["attn"], atol=1e-4) and torch.allclose(feat_s[1].cpu(), res["src"]["feat"][1], atol=1e-4)
    loss = Fn.cross_entropy(out_s, ys.cuda())
    for l in range(3):
        if place_adv[l] == "Y":
            ps, pt = pd_s[l].reshape(-1, 2), pd_t[l].reshape(-1, 2)
            lab = torch.cat((torch.zeros(ps.size(0)), torch.ones(pt.size(0)))).long().cuda()
            loss = loss + Fn.cross_entropy(torch.cat((ps, pt)), lab)
    loss.backward()
    assert abs(loss.item() - res["loss"].item()) < 2e-4
    named = dict(m.named_parameters())
    assert {k for k, v in named.items() if v.grad is not None} == set(res["grads"])
    for k, w in res["grads"].items():
        got = named[k].grad.cpu()
        assert torch.allclose(got, w, rtol=2e-3, atol=2e-4 * w.abs().max().item() + 1e-7), (k, (got - w).abs().max().item(), w.abs().max().item())


@pytest.mark.gpu
def test_module_path_avgpool_validates_with_another_segment_count():
    """models.py:60, 555: val_segments (default 25) need not equal train_segments; TemPooling averages whatever it is given
    (models.py:421-433).  The module path builds a plan per segment count: a train-mode pass on 4 segments, then an eval-mode pass
    on 9 (main.validate feeds the validation clip in BOTH slots, main.py:707), both against the oracle."""
    from ta3n_amd.models import VideoModel
    C_, Tt, Tv, D, Fc, Bs, Bt = 7, 4, 9, 512, 64, 6, 5
    mk = lambda T: orc.Config(num_class=C_, num_segments=T, feature_dim=D, fc_dim=Fc, dropout_i=0.0, dropout_v=0.0, place_adv=("N", "Y", "Y"),
                              add_loss_DA="none", use_attn="none", frame_aggregation="avgpool")
    params = synth_state(orc.param_shapes(mk(Tt)), seed=21)
    assert orc.param_shapes(mk(Tv)) == orc.param_shapes(mk(Tt))
    m = VideoModel(C_, "video", "avgpool", "RGB", train_segments=Tt, val_segments=Tv, base_model="resnet18", fc_dim=Fc, dropout_i=0.0,
                   dropout_v=0.0, use_attn="none", verbose=False)
    sd = m.state_dict(); sd.update(params); m.load_state_dict(sd)
    m = m.cuda()
    m.train()
    xs, xt, ys, yt = synth_batch(C_, Tt, D, Bs, Bt, seed=22)
    out = m(xs, xt, BETA, 0, True, False)
    with torch.no_grad():
        want = orc.forward_domain(params, xs, BETA, mk(Tt), domain="S")
    assert torch.allclose(out[1].cpu(), want["out"], atol=1e-3)
    m.eval()
    xv, _, _, _ = synth_batch(C_, Tv, D, Bs, Bt, seed=23)
    with torch.no_grad():
        ev = m(xv, xv, [0, 0, 0], 0, False, False)
        want = orc.forward_domain(params, xv, [0, 0, 0], mk(Tv), domain="T")
    assert ev[6].shape == (Bs, C_) and ev[9][2].shape == (Bs, Tv, Fc)
    assert torch.allclose(ev[6].cpu(), want["out"], atol=1e-3) and torch.allclose(ev[9][1].cpu(), want["feat"][1], atol=1e-4)
    with pytest.raises(ValueError):      # trn-m: the relation module is built for train_segments frames, in the reference too
        VideoModel(C_, "video", "trn-m", "RGB", train_segments=Tt, val_segments=Tv, base_model="resnet18", fc_dim=Fc, verbose=False).cuda()(
            xv, xv, [0, 0, 0], 0, False, False)
