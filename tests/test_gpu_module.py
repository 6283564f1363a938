"""GPU: the drop-in module path.  ta3n_amd.models.VideoModel is driven the way the
reference's main.train drives models.VideoModel (main.py:418-583): forward, the loss
assembled OUTSIDE the model from its outputs, autograd backward, clip_grad_norm_,
torch.optim.SGD(nesterov) - and must land on the parameters the reference itself produced."""
import pytest
import torch

from golden_util import Golden, case_config, step_schedule
from oracle import ta3n_oracle as orc
from ta3n_amd.synthetic import synth_batch, synth_state

pytestmark = pytest.mark.gpu


def _model(c):
    from ta3n_amd.models import VideoModel
    arch = "resnet18" if c["D"] == 512 else "resnet101"
    m = VideoModel(c["C"], "video", "trn-m", "RGB", train_segments=c["T"], val_segments=c["T"], base_model=arch,
                   fc_dim=c["fc_dim"], dropout_i=0.0, dropout_v=0.0, partial_bn=False, verbose=False)
    sd = m.state_dict()
    sd.update(synth_state({k: tuple(v.shape) for k, v in sd.items()}, seed=c["wseed"], scale=c["wscale"]))
    m.load_state_dict(sd)
    return m.cuda()


@pytest.mark.parametrize("name", ["tiny_T5", "tiny_T9", "headline"])
def test_module_autograd_path_matches_reference(name):
    g = Golden(name)
    c = case_config(g)
    model = _model(c)
    model.train()
    opt = torch.optim.SGD(model.parameters(), c["lr"], momentum=0.9, weight_decay=1e-4, nesterov=True)
    cfg = orc.Config(num_class=c["C"], num_segments=c["T"], feature_dim=c["D"], fc_dim=c["fc_dim"])
    live = set(str(k) for k in g.meta("live"))
    for s, st in enumerate(step_schedule(c)):
        xs, xt, ys, yt = synth_batch(c["C"], c["T"], c["D"], c["Bs"], c["Bt"], seed=st["xseed"])
        xs[st["n_src"]:] = 0; xt[st["n_tgt"]:] = 0
        for pg in opt.param_groups:
            pg["lr"] = st["lr"]
        out = model(xs, xt, [0.75, 0.75, 0.5], 0, True, False)          # CPU inputs, like main.py:418
        attn_s, out_s, out_s2, pd_s, feat_s, attn_t, out_t, out_t2, pd_t, feat_t = out
        if s == 0:
            g.check("fwd/out_s", out_s, 0, 1e-3); g.check("fwd/out_t", out_t, 0, 1e-3)
            g.check("fwd/attn_s", attn_s, 2e-4, 5e-5)
            for i, nm in enumerate(("rel", "vid", "frm")):
                g.check(f"fwd/pd_s_{nm}", pd_s[i], 0, 1e-3); g.check(f"fwd/pd_t_{nm}", pd_t[i], 0, 1e-3)
            g.check("fwd/feat_s_v", feat_s[1], 2e-4, 5e-5); g.check("fwd/feat_t_f1", feat_t[2], 2e-4, 5e-5)
        src = dict(out=out_s, pred_domain=pd_s)
        tgt = dict(out=out_t, pred_domain=pd_t)
        loss, _ = orc.total_loss(src, tgt, ys.cuda(), 0.003, cfg, st["n_src"], st["n_tgt"])   # main.py:439-562
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), c["clip"])
        for k, p in model.named_parameters():
            assert (p.grad is not None) == (k in live), k
            if k in live:
                g.check(f"step{s}/clipped_grad/{k}", p.grad, 1e-3, 2e-5, rms_atol=1e-2 if s == 0 else 0.15)
        opt.step()
        for k, p in model.named_parameters():
            g.check(f"step{s}/param/{k}", p, 2e-4, 5e-5)


def test_module_eval_mode_and_state_dict_roundtrip():
    g = Golden("tiny_T5")
    c = case_config(g)
    model = _model(c)
    model.eval()
    xs, xt, _, _ = synth_batch(c["C"], c["T"], c["D"], c["Bs"], c["Bt"], seed=c["xseed"])
    with torch.no_grad():                                              # main.validate: same data in both slots, beta 0
        o = model(xs, xs, [0, 0, 0], 0, False, False)
    g.check("fwd/out_s", o[1], 0, 1e-3)
    assert torch.equal(o[1], o[6])
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    from ta3n_amd.models import VideoModel
    m2 = VideoModel(c["C"], "video", "trn-m", "RGB", train_segments=c["T"], val_segments=c["T"], base_model="resnet18",
                    fc_dim=c["fc_dim"], dropout_i=0.0, dropout_v=0.0, verbose=False)
    m2.load_state_dict(sd)
    m2 = m2.cuda().eval()
    with torch.no_grad():
        o2 = m2(xs, xs, [0, 0, 0], 0, False, False)
    assert torch.equal(o[1], o2[1])


def test_validation_metrics_match_reference_accuracy():
    """main.validate's bookkeeping (CE, accuracy() top-1/top-5 of main.py:809-822, confusion matrix of
    test_models.py:198) accumulated on the device over several ragged batches vs torch on the CPU."""
    from ta3n_amd.engine import TrainEngine
    C_, T, D, Fc = 12, 5, 512, 64
    eng = TrainEngine(16, 4, T, D, Fc, C_, dropout_i=0.5, dropout_v=0.5)
    cfg = orc.Config(num_class=C_, num_segments=T, feature_dim=D, fc_dim=Fc)
    params = synth_state(orc.param_shapes(cfg), seed=4)
    eng.load_state(params)
    g = torch.Generator().manual_seed(8)
    logits, labels = [], []
    for i, n in enumerate((16, 7, 1, 12)):
        x = torch.randn(n, T, D, generator=g).abs()
        y = torch.randint(0, C_, (n,), generator=g)
        eng.evaluate_batch(x.cuda(), y.cuda(), reset=(i == 0))
        with torch.no_grad():
            ref = orc.forward_domain({k: v for k, v in params.items()}, x, [0.0, 0.0, 0.0], cfg)
        logits.append(ref["out"]); labels.append(y)
    res = eng.eval_results()
    out, lab = torch.cat(logits), torch.cat(labels)
    assert res["n"] == 36
    assert abs(res["loss"] - torch.nn.functional.cross_entropy(out, lab).item()) < 1e-4
    top5 = out.topk(5, 1).indices
    assert abs(res["prec1"] - 100.0 * (top5[:, 0] == lab).float().mean().item()) < 1e-4
    assert abs(res["prec5"] - 100.0 * (top5 == lab[:, None]).any(1).float().mean().item()) < 1e-4
    conf = torch.zeros(C_, C_, dtype=torch.int32)
    for p_, l_ in zip(out.argmax(1).tolist(), lab.tolist()):
        conf[l_, p_] += 1
    assert torch.equal(res["confusion"], conf)


def test_standalone_relation_module_forward_matches_reference_math():
    """TRNmodule.RelationModuleMultiScale.forward on its own (north_star names it first): against the reference's formula
    (TRNmodule.py:58-82) written with torch ops on the CPU - ReLU, gather + concat of each selected tuple, Linear, ReLU, per-scale sum."""
    import torch.nn.functional as Fn
    from ta3n_amd.TRNmodule import RelationModuleMultiScale
    torch.manual_seed(3)
    for T, D, B in ((5, 128, 7), (9, 64, 4), (3, 512, 33)):
        m = RelationModuleMultiScale(D, 256, T, verbose=False)
        x = torch.randn(B, T, D)
        want = []
        for j, tuples in enumerate(m.relations_selected):
            lin = m.fc_fusion_scales[j][1]
            acc = 0
            for tup in tuples:
                a = torch.relu(x[:, list(tup), :]).reshape(B, -1)
                acc = acc + torch.relu(Fn.linear(a, lin.weight, lin.bias))
            want.append(acc.unsqueeze(1))
        want = torch.cat(want, 1).detach()
        with torch.no_grad():
            got = m(x.cuda()).cpu()
        assert got.shape == (B, T - 1, 256)
        assert torch.allclose(got, want, rtol=2e-4, atol=2e-5), (got - want).abs().max()


def test_standalone_relation_module_backward_matches_autograd_of_the_reference_math():
    """... and its backward (TRNmodule.py:58-82 is differentiable): gradients at the input and at every fusion layer's weight / bias
    through the plan's TRN weight-gradient / input-gradient launch, against torch autograd of the same formula on the CPU."""
    import torch.nn.functional as Fn
    from ta3n_amd.TRNmodule import RelationModuleMultiScale
    torch.manual_seed(5)
    for T, D, B in ((5, 128, 7), (9, 64, 4), (3, 512, 33)):
        m = RelationModuleMultiScale(D, 256, T, verbose=False)
        x = torch.randn(B, T, D)
        xr = x.clone().requires_grad_(True)
        ref_params = [p.detach().clone().requires_grad_(True) for p in m.parameters()]
        want = []
        for j, tuples in enumerate(m.relations_selected):
            w, b = ref_params[2 * j], ref_params[2 * j + 1]
            acc = 0
            for tup in tuples:
                a = torch.relu(xr[:, list(tup), :]).reshape(B, -1)
                acc = acc + torch.relu(Fn.linear(a, w, b))
            want.append(acc.unsqueeze(1))
        want = torch.cat(want, 1)
        gout = torch.randn_like(want)
        want.backward(gout)
        m = m.cuda()
        xg = x.cuda().requires_grad_(True)
        got = m(xg)
        got.backward(gout.cuda())
        assert torch.allclose(got.detach().cpu(), want.detach(), rtol=2e-4, atol=2e-5)
        assert torch.allclose(xg.grad.cpu(), xr.grad, rtol=2e-4, atol=2e-5 * xr.grad.abs().max().item()), (xg.grad.cpu() - xr.grad).abs().max()
        for p, r in zip(m.parameters(), ref_params):
            assert torch.allclose(p.grad.cpu(), r.grad, rtol=2e-4, atol=2e-5 * r.grad.abs().max().item()), (p.grad.cpu() - r.grad).abs().max()


def test_standalone_relation_module_called_twice_before_one_backward():
    """The reference calls the module for the source and for the target batch and backpropagates once (models.py:636-651).  With
    EQUAL batch sizes both calls share the module's cached workspace: each call must keep what its own backward needs."""
    import torch.nn.functional as Fn
    from ta3n_amd.TRNmodule import RelationModuleMultiScale
    torch.manual_seed(9)
    T, D, B = 5, 128, 6
    m = RelationModuleMultiScale(D, 256, T, verbose=False)
    xs, xt = torch.randn(B, T, D), torch.randn(B, T, D)
    ref_params = [p.detach().clone().requires_grad_(True) for p in m.parameters()]
    xr = [xs.clone().requires_grad_(True), xt.clone().requires_grad_(True)]

    def ref(x):
        out = []
        for j, tuples in enumerate(m.relations_selected):
            w, b = ref_params[2 * j], ref_params[2 * j + 1]
            acc = 0
            for tup in tuples:
                acc = acc + torch.relu(Fn.linear(torch.relu(x[:, list(tup), :]).reshape(B, -1), w, b))
            out.append(acc.unsqueeze(1))
        return torch.cat(out, 1)
    gs, gt = torch.randn(B, T - 1, 256), torch.randn(B, T - 1, 256)
    ((ref(xr[0]) * gs).sum() + (ref(xr[1]) * gt).sum()).backward()
    m = m.cuda()
    xg = [xs.cuda().requires_grad_(True), xt.cuda().requires_grad_(True)]
    os_, ot = m(xg[0]), m(xg[1])                       # two forwards ...
    ((os_ * gs.cuda()).sum() + (ot * gt.cuda()).sum()).backward()      # ... one backward
    for a, b in zip(xg, xr):
        assert torch.allclose(a.grad.cpu(), b.grad, rtol=2e-4, atol=2e-5 * b.grad.abs().max().item()), (a.grad.cpu() - b.grad).abs().max()
    for p, r in zip(m.parameters(), ref_params):
        assert torch.allclose(p.grad.cpu(), r.grad, rtol=2e-4, atol=2e-5 * r.grad.abs().max().item()), (p.grad.cpu() - r.grad).abs().max()
