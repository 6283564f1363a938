"""dis_DA DAN / JAN and ens_DA MCD on the module path, end to end on the GPU: this repository's main.train
(the reference's loss assembly, main.py:439-562) on ta3n_amd.models.VideoModel (HIP forward / backward through the C ABI) with
ta3n_amd.loss, against fixtures written by the reference's own main.train with the same options
(tests/golden/make_golden.py: tiny_dan, tiny_dan_all, tiny_jan, tiny_mcd, mid_dan_mcd): clipped gradients, parameters after
each step, and the loss_d / loss_s the reference logged."""
import argparse
import importlib.util
import io
import os
import re

import numpy as np
import pytest
import torch

from golden_util import AVG_DA_EXTRA_CASES, BN_CASES, DA_EXTRA_CASES, Golden, case_config, step_schedule
from ta3n_amd.synthetic import synth_batch, synth_state

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_main():
    spec = importlib.util.spec_from_file_location("ta3n_repo_main", os.path.join(ROOT, "main.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _args(c):
    a = argparse.Namespace()
    a.no_partialbn = True
    a.batch_size = [c["Bs"], c["Bt"], c["Bs"]]
    a.baseline_type, a.num_segments, a.pretrain_source, a.pred_normalize, a.tensorboard = "video", c["T"], False, "N", False
    a.use_target, a.adv_DA, a.place_adv = "uSv", "RevGrad", ["Y", "Y", "Y"]
    a.add_loss_DA, a.use_attn = "attentive_entropy", "TransAttn"
    if c["agg"] == "avgpool":        # TemPooling + RevGrad on the fixture's levels, no attention (make_golden.make_args)
        a.place_adv, a.add_loss_DA, a.use_attn = list(c["place_adv"]), "none", "none"
    if c.get("add_loss_DA"):
        a.add_loss_DA = c["add_loss_DA"]
    a.dis_DA, a.place_dis, a.ens_DA = c["dis_DA"], list(c["place_dis"]), c["ens_DA"]
    a.clip_gradient, a.verbose, a.print_freq, a.show_freq = c["clip"], False, 1, 10 ** 9
    a.lr_adaptive, a.lr, a.save_attention, a.epochs, a.add_fc = "dann", c["lr"], -1, 30, 1
    a.momentum, a.weight_decay = 0.9, 1e-4
    return a


class _FakeDP:      # main.train uses model.module, model.train(), model(...), model.parameters()  (main.py:79)
    def __init__(self, m):
        self.module = m

    def __call__(self, *a, **k):
        return self.module(*a, **k)

    def train(self, mode=True):
        return self.module.train(mode)

    def parameters(self):
        return self.module.parameters()


@pytest.mark.parametrize("name", DA_EXTRA_CASES + BN_CASES + AVG_DA_EXTRA_CASES)
def test_main_train_with_discrepancy_and_ensemble_losses_matches_the_reference(name):
    from ta3n_amd.models import VideoModel
    main = _load_main()
    g = Golden(name)
    c = case_config(g)
    T, C = c["T"], c["C"]
    arch = str(g.meta("arch"))
    avg = c["agg"] == "avgpool"
    model = VideoModel(C, "video", "avgpool" if avg else "trn-m", "RGB", train_segments=T, val_segments=T, base_model=arch, add_fc=1,
                       fc_dim=c["fc_dim"], dropout_i=0.0, dropout_v=0.0, partial_bn=False, use_bn=c["use_bn"], ens_DA=c["ens_DA"],
                       use_attn="none" if avg else "TransAttn", verbose=False).cuda()
    sd = model.state_dict()
    shapes = {k: tuple(v.shape) for k, v in sd.items()}
    sd.update({k: v.cuda() for k, v in synth_state(shapes, seed=c["wseed"], scale=c["wscale"]).items()})
    model.load_state_dict(sd)
    assert ("fc_classifier_video_source_2.weight" in shapes) == (c["ens_DA"] == "MCD")
    args = _args(c)
    main.args = args
    opt = torch.optim.SGD(model.parameters(), args.lr, momentum=args.momentum, weight_decay=args.weight_decay, nesterov=True)
    crit, crit_d = torch.nn.CrossEntropyLoss().cuda(), torch.nn.CrossEntropyLoss().cuda()
    wrapped = _FakeDP(model)
    n_steps = c["steps"]
    log, log_short = io.StringIO(), io.StringIO()
    live = set(str(k) for k in g.meta("live"))
    bn = c["use_bn"] != "none"
    xs0, xt0, _, _ = synth_batch(C, T, c["D"], c["Bs"], c["Bt"], seed=c["xseed"])
    if bn:      # the fixture's plain train-mode forward comes first and moves the BatchNorm buffers like any train-mode pass
        model.train()
        with torch.no_grad():
            o = model(xs0.cuda(), xt0.cuda(), [0.75, 0.75, 0.5], 0, True, False)
        g.check("fwd/out_s", o[1], 2e-4, 2e-4)
        g.check("fwd/feat_t_v", o[9][1], 2e-4, 2e-4)
    for s, st in enumerate(step_schedule(c)):
        xs, xt, ys, yt = synth_batch(C, T, c["D"], c["Bs"], c["Bt"], seed=st["xseed"])
        if st["n_src"] < c["Bs"] or st["n_tgt"] < c["Bt"]:       # a short last batch: main.train pads it back with zero rows (main.py:359-364)
            xs, ys, xt, yt = xs[:st["n_src"]], ys[:st["n_src"]], xt[:st["n_tgt"]], yt[:st["n_tgt"]]
        args.epochs = 30 * n_steps
        main.train(C, [(xs, ys)], [(xt, yt)], wrapped, crit, crit_d, opt, n_steps + s, log, log_short, c["alpha"],
                   [0.75, 0.75, 0.5], 0.0 if avg else 0.003, c["mu"])
        torch.cuda.synchronize()
        got_live = {k for k, v in model.named_parameters() if v.grad is not None}
        assert got_live == live, (got_live ^ live)
        for k, v in model.named_parameters():
            if k in live:
                g.check(f"step{s}/clipped_grad/{k}", v.grad, 2e-4, 5e-6, rms_atol=2e-4)
            g.check(f"step{s}/param/{k}", v, 2e-4, 5e-6)
    if bn:      # buffers after one plain + n_steps train-mode passes, then main.validate's eval-mode forward through them
        sd = model.state_dict()
        for d in "ST":
            g.check(f"final/state/bn_shared_{d}.running_mean", sd[f"bn_shared_{d}.running_mean"], 2e-4, 2e-5)
            g.check(f"final/state/bn_shared_{d}.running_var", sd[f"bn_shared_{d}.running_var"], 2e-4, 2e-5)
            assert int(sd[f"bn_shared_{d}.num_batches_tracked"]) == int(g.z[f"final/state/bn_shared_{d}.num_batches_tracked#full"])
        model.eval()
        with torch.no_grad():
            ev = model(xs0.cuda(), xs0.cuda(), [0, 0, 0], 0, False, False)
        g.check("eval/out_t", ev[6], 3e-4, 3e-4)
        g.check("eval/feat_t_v", ev[9][1], 3e-4, 3e-4)
        model.train()
    # the loss components the reference logged (running averages over the steps of one "epoch" = one step each here)
    want, got = str(g.meta("log")).strip().splitlines(), log.getvalue().strip().splitlines()
    assert len(want) == len(got) == n_steps
    for w, h in zip(want, got):
        for key in ("Loss", "loss_c", "loss_d", "loss_a", "loss_e", "loss_s"):
            mw = re.search(key + r" (-?[0-9.]+)", w)
            mh = re.search(key + r" (-?[0-9.]+)", h)
            assert (mw is None) == (mh is None), (key, w, h)
            if mw:
                assert abs(float(mw.group(1)) - float(mh.group(1))) <= 2e-3 * max(1.0, abs(float(mw.group(1)))), (key, w, h)


def test_loss_functions_match_the_oracle_restatement_and_its_gradients():
    """ta3n_amd.loss.mmd_rbf / JAN (one matrix product for the pairwise distances) against the oracle's broadcast form of
    loss.py:46-120, values and gradients."""
    from oracle import ta3n_oracle as orc
    from ta3n_amd import loss as L
    torch.manual_seed(3)
    for n, d in ((4, 12), (37, 256), (128, 256)):
        a = torch.randn(n, d, dtype=torch.float64).cuda().requires_grad_(True)
        b = (0.5 * torch.randn(n, d, dtype=torch.float64) + 0.3).cuda().requires_grad_(True)
        a2, b2 = a.detach().cpu().requires_grad_(True), b.detach().cpu().requires_grad_(True)
        for num in (2, 5):
            v = L.mmd_rbf(a, b, kernel_mul=2.0, kernel_num=num)
            w = orc.mmd_rbf(a2, b2, 2.0, num)
            assert abs(v.item() - w.item()) < 1e-9 * max(1.0, abs(w.item()))
            ga, = torch.autograd.grad(v, a, retain_graph=True)
            gw, = torch.autograd.grad(w, a2, retain_graph=True)
            assert torch.allclose(ga.cpu(), gw, rtol=1e-7, atol=1e-12)
        y1 = torch.randn(n, 7, dtype=torch.float64).cuda().requires_grad_(True)
        y2 = torch.randn(n, 7, dtype=torch.float64).cuda().requires_grad_(True)
        v = L.JAN([y1, a], [y2, b])
        w = orc.jan([y1.detach().cpu(), a2], [y2.detach().cpu(), b2])
        assert abs(v.item() - w.item()) < 1e-9 * max(1.0, abs(w.item()))


@pytest.mark.parametrize("n,d", [(4, 12), (37, 256), (74, 256), (128, 12), (200, 512)])
def test_hip_kernel_matrix_and_its_gradient_match_the_oracle(n, d):
    """fp32 CUDA tensors take the HIP path of ta3n_amd.loss (ta3n_gaussian_kernel / ta3n_mmd_rowdiff, csrc/ta3n_mmd.hip): mmd_rbf and
    JAN values and feature gradients against the oracle's fp64 restatement of loss.py:46-120 - ragged sizes, fixed and
    data-dependent bandwidths, duplicated rows (exact zeros on the diagonal: the explicit-difference form)."""
    from oracle import ta3n_oracle as orc
    from ta3n_amd import loss as L
    g = torch.Generator().manual_seed(n + d)
    a64 = torch.randn(n, d, generator=g, dtype=torch.float64)
    b64 = 0.5 * torch.randn(n, d, generator=g, dtype=torch.float64) + 0.3
    b64[: n // 4] = a64[: n // 4]                                    # duplicated rows across the domains
    a = a64.float().cuda().requires_grad_(True)
    b = b64.float().cuda().requires_grad_(True)
    a2, b2 = a64.float().double().requires_grad_(True), b64.float().double().requires_grad_(True)
    for num, sigma in ((5, None), (2, None), (5, 1.68)):
        v = L.mmd_rbf(a, b, kernel_mul=2.0, kernel_num=num, fix_sigma=sigma)
        w = orc.mmd_rbf(a2, b2, 2.0, num) if sigma is None else None
        if w is None:      # the oracle's signature has no fixed bandwidth: the torch form of the same module in fp64
            w = L.mmd_rbf(a2, b2, kernel_mul=2.0, kernel_num=num, fix_sigma=sigma)
        assert abs(v.item() - w.item()) <= 2e-5 * max(1.0, abs(w.item())), (num, sigma, v.item(), w.item())
        ga, gb = torch.autograd.grad(v, (a, b))
        wa, wb = torch.autograd.grad(w, (a2, b2))
        for got, want in ((ga, wa), (gb, wb)):      # fp32 kernels vs fp64: relative L2 per tensor (single entries are differences of nearly equal terms)
            err = (got.cpu().double() - want).norm().item()      # (a fixed bandwidth far below the distances underflows both sides to ~0)
            assert err <= 5e-4 * want.norm().item() + 1e-9, (num, sigma, err, want.norm().item())
    y1 = torch.randn(n, 7, generator=g).cuda().requires_grad_(True)
    y2 = torch.randn(n, 7, generator=g).cuda().requires_grad_(True)
    v = L.JAN([y1, a], [y2, b])
    y1d, y2d = y1.detach().cpu().double().requires_grad_(True), y2.detach().cpu().double().requires_grad_(True)
    w = orc.jan([y1d, a2], [y2d, b2])
    assert abs(v.item() - w.item()) <= 2e-5 * max(1.0, abs(w.item()))
    got = torch.autograd.grad(v, (y1, a, b))
    want = torch.autograd.grad(w, (y1d, a2, b2))
    for x, y in zip(got, want):
        err = (x.cpu().double() - y).norm().item()
        assert err <= 5e-4 * y.norm().item() + 1e-9, (err, y.norm().item())
    # bitwise reproducible (fixed-order partial sums)
    v2 = L.mmd_rbf(a, b)
    assert torch.equal(v2, L.mmd_rbf(a, b))


@pytest.mark.parametrize("name", ["tiny_dan", "tiny_dan_all", "tiny_jan"])
def test_engine_path_with_discrepancy_losses_matches_the_reference(name):
    """TrainEngine(dis_DA=...): the same fixtures as above (written by the reference's main.train with --dis_DA DAN / JAN) through the
    ENGINE - ta3n_forward, ta3n_loss, the HIP discrepancy kernels, ta3n_backward, ta3n_sgd_step - instead of the module path:
    clipped gradients, parameters after each step, the logged loss_d."""
    from ta3n_amd.engine import TrainEngine
    g = Golden(name)
    c = case_config(g)
    T, C = c["T"], c["C"]
    eng = TrainEngine(c["Bs"], c["Bt"], T, c["D"], c["fc_dim"], C, dropout_i=0.0, dropout_v=0.0, clip=c["clip"], dis_DA=c["dis_DA"],
                      place_dis=c["place_dis"], alpha=c["alpha"])
    assert not eng.fused
    eng.load_state(synth_state({n: s for n, _, s, _ in eng.plan.params}, seed=c["wseed"], scale=c["wscale"]))
    live = set(str(k) for k in g.meta("live"))
    assert set(eng.live_names()) == live
    want_log = str(g.meta("log")).strip().splitlines()
    for s, st in enumerate(step_schedule(c)):
        xs, xt, ys, yt = synth_batch(C, T, c["D"], c["Bs"], c["Bt"], seed=st["xseed"])
        eng.set_batch(xs.cuda(), xt.cuda(), ys.cuda())
        eng.train_step([0.75, 0.75, 0.5], 0.003, st["lr"], valid_source=st["n_src"], valid_target=st["n_tgt"])
        torch.cuda.synchronize()
        coef = eng.region("grad_norm")[1].item()
        raw = eng.param_views(eng.G)
        for k, v in eng.param_views().items():
            if k in live:
                g.check(f"step{s}/clipped_grad/{k}", raw[k].cpu() * coef, 2e-4, 5e-6, rms_atol=2e-4)
            g.check(f"step{s}/param/{k}", v.cpu(), 2e-4, 5e-6)
        m = re.search(r"loss_d (-?[0-9.]+)", want_log[s])
        assert m is not None and abs(float(m.group(1)) - eng.loss_d.item()) <= 2e-3 * max(1.0, abs(float(m.group(1))))


@pytest.mark.parametrize("dis_DA,place,ns,nt", [("DAN", ("Y", "Y", "N"), 40, 33), ("DAN", ("N", "Y", "N"), 37, 37), ("DAN", ("Y", "N", "N"), 5, 40),
                                               ("JAN", ("Y", "Y", "N"), 40, 29), ("DAN", ("Y", "Y", "N"), 512, 512), ("DAN", ("Y", "Y", "N"), 0, 12),
                                               ("DAN", ("Y", "Y", "N"), 128, 74), ("JAN", ("Y", "Y", "N"), 128, 74)])      # (the last two: BASELINE configs[1]'s full shape)
def test_native_discrepancy_matches_the_autograd_path(dis_DA, place, ns, nt, monkeypatch):
    """ta3n_discrepancy (one rank: the whole DAN / JAN term from the library) against the torch-autograd assembly around the same kernels
    (parallel.discrepancy_over_ranks, what more than one rank still runs): loss, the gradient added to the logit gradient and the one written
    to the feature-gradient entry - ragged valid counts, one feature only, two chunks of 256 videos, an empty domain."""
    from ta3n_amd.engine import TrainEngine
    Bs, Bt = max(ns, 8), max(nt, 8)
    T, D, F, C = (5, 2048, 512, 12) if (ns, nt) == (128, 74) else (3, 64, 64, 7)
    out = {}
    for native in ("1", "0"):
        monkeypatch.setenv("TA3N_NATIVE_DISCREPANCY", native)
        eng = TrainEngine(Bs, Bt, T, D, F, C, dropout_i=0.0, dropout_v=0.0, clip=20.0, dis_DA=dis_DA, place_dis=place, alpha=0.7)
        eng.load_state(synth_state({n: s for n, _, s, _ in eng.plan.params}, seed=5, scale="trained"))
        xs, xt, ys, yt = synth_batch(C, T, D, Bs, Bt, seed=9)
        eng.set_batch(xs.cuda(), xt.cuda(), ys.cuda())
        eng.set_hyper([0.75, 0.75, 0.5], 0.003, 1e-3, train=True, valid_source=ns, valid_target=nt)
        eng.forward(); eng.loss()
        gy0 = eng.region("gY", (Bs + Bt, C)).clone()
        eng.discrepancy()
        torch.cuda.synchronize()
        assert (eng._disc_scratch is not None) == (native == "1")
        out[native] = (float(eng.loss_d), (eng.region("gY", (Bs + Bt, C)) - gy0).cpu(), eng.region("gV_ext", (Bs + Bt, -1)).clone().cpu())
    (l1, gy1, gv1), (l0, gy0_, gv0) = out["1"], out["0"]
    assert abs(l1 - l0) <= 1e-5 * max(1.0, abs(l0)), (l1, l0)
    for a, b in ((gy1, gy0_), (gv1, gv0)):
        assert (a - b).norm().item() <= 1e-5 * b.norm().item() + 1e-12, ((a - b).norm().item(), b.norm().item())
    if min(ns, nt) > 0:
        assert gv0.abs().sum() > 0 or place[1] != "Y"
        assert gv1[min(ns, nt):Bs].abs().sum() == 0          # videos beyond min(#source, #target) take no part
