"""ta3n_amd.accel: torch.nn.utils.clip_grad_norm_ and torch.optim.SGD.step as passes over VideoModel's flat parameter / gradient
buffers (what compat/ installs under the reference's main.py) - the same numbers as torch's per-tensor code, the optimiser's
state_dict still the reference checkpoint's, and torch's own code whenever the preconditions do not hold."""
import copy

import pytest
import torch
import torch.nn.functional as F

from ta3n_amd import accel
from ta3n_amd.loss import attentive_entropy
from ta3n_amd.models import VideoModel
from ta3n_amd.synthetic import synth_batch

pytestmark = pytest.mark.gpu

Bs, Bt, T, D, C = 6, 4, 5, 512, 12


def _model(seed=0, **kw):
    torch.manual_seed(seed)
    m = VideoModel(C, "video", "trn-m", "RGB", train_segments=T, val_segments=T, base_model="resnet18", fc_dim=64, dropout_i=0.0,
                   dropout_v=0.0, verbose=False, **kw).cuda()
    m.train()
    return m


def _step(m, opt, seed, max_norm, levels=(0, 1, 2)):
    xs, xt, ys, yt = synth_batch(C, T, D, Bs, Bt, seed=seed)
    o = m(xs.cuda(), xt.cuda(), [0.75, 0.75, 0.5], 0, True, False)
    loss = F.cross_entropy(o[1], ys.cuda())
    pd_all = {}
    for l in levels:
        ps, pt = o[3][l].reshape(-1, 2), o[8][l].reshape(-1, 2)
        lab = torch.cat((torch.zeros(ps.size(0)), torch.ones(pt.size(0)))).long().cuda()
        pd_all[l] = torch.cat((ps, pt))
        loss = loss + F.cross_entropy(pd_all[l], lab)
    if 1 in pd_all:
        loss = loss + 0.003 * attentive_entropy(torch.cat((o[1], o[6])), pd_all[1])
    opt.zero_grad()
    loss.backward()
    total = torch.nn.utils.clip_grad_norm_(m.parameters(), max_norm)
    opt.step()
    return float(total)


@pytest.mark.parametrize("max_norm", [1e9, 0.05])
def test_flat_clip_and_step_give_torchs_numbers(max_norm):
    ref = _model()
    fast = copy.deepcopy(ref)
    o_ref = torch.optim.SGD(ref.parameters(), 3e-2, momentum=0.9, weight_decay=1e-4, nesterov=True)
    o_fast = torch.optim.SGD(fast.parameters(), 3e-2, momentum=0.9, weight_decay=1e-4, nesterov=True)
    try:
        for s in range(4):
            accel.uninstall()
            n_ref = _step(ref, o_ref, 10 + s, max_norm)
            assert accel.install()
            n_fast = _step(fast, o_fast, 10 + s, max_norm)
            assert abs(n_ref - n_fast) <= 1e-5 * n_ref
            assert fast._mom_flat is not None and fast._grad_flat is not None          # the flat paths ran
            for (k, a), (_, b) in zip(ref.named_parameters(), fast.named_parameters()):  # ... and left the clipped gradients in place
                assert (a.grad is None) == (b.grad is None), k
                if a.grad is not None:
                    assert torch.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-8), k
        pr, pf = dict(ref.named_parameters()), dict(fast.named_parameters())
        for k in pr:
            if max_norm > 1e6:      # no clipping: the same operations element for element
                assert torch.equal(pr[k], pf[k]), k
            else:                   # the clip coefficient comes from one flat reduction instead of a norm of norms
                assert torch.allclose(pr[k], pf[k], rtol=1e-5, atol=1e-7), k
        # the optimiser's momentum buffers are views into ONE buffer and carry torch's values
        base = fast._mom_flat.data_ptr()
        n_views = 0
        for k, p in pf.items():
            st = o_fast.state.get(p, {})
            if "momentum_buffer" in st:
                n_views += 1
                assert base <= st["momentum_buffer"].data_ptr() < base + 4 * fast._mom_flat.numel()
                ref_buf = o_ref.state[pr[k]]["momentum_buffer"]
                assert torch.allclose(st["momentum_buffer"], ref_buf, rtol=1e-5, atol=1e-7) if max_norm < 1e6 else torch.equal(st["momentum_buffer"], ref_buf), k
        assert n_views == sum(1 for p in pr.values() if p in o_ref.state)
        # a checkpoint round trip (main.py:266-274, 94-106) continues bit-identically
        sd_m, sd_o = copy.deepcopy(fast.state_dict()), copy.deepcopy(o_fast.state_dict())
        again = _model(seed=5)
        again.load_state_dict(sd_m)
        o_again = torch.optim.SGD(again.parameters(), 1e-3, momentum=0.9, weight_decay=1e-4, nesterov=True)
        o_again.load_state_dict(sd_o)
        _step(fast, o_fast, 99, max_norm)
        _step(again, o_again, 99, max_norm)
        for (k, a), (_, b) in zip(fast.named_parameters(), again.named_parameters()):
            assert torch.equal(a, b), k
    finally:
        accel.uninstall()


def test_torchs_own_code_runs_when_the_preconditions_do_not_hold():
    try:
        assert accel.install()
        # (a) a discriminator without a loss: its parameters keep grad None, torch skips them (no weight decay either)
        m = _model()
        before = {k: v.detach().clone() for k, v in m.named_parameters()}
        opt = torch.optim.SGD(m.parameters(), 3e-2, momentum=0.9, weight_decay=1e-4, nesterov=True)
        _step(m, opt, 3, 20.0, levels=(0, 1))
        after = dict(m.named_parameters())
        assert torch.equal(before["fc_feature_domain.weight"], after["fc_feature_domain.weight"])          # frame discriminator untouched
        assert not torch.equal(before["fc_feature_shared_source.weight"], after["fc_feature_shared_source.weight"])
        # (b) another optimiser, another module
        lin = torch.nn.Linear(8, 4).cuda()
        o2 = torch.optim.Adam(lin.parameters(), 1e-2)
        lin(torch.randn(3, 8, device="cuda")).sum().backward()
        n = torch.nn.utils.clip_grad_norm_(lin.parameters(), 1.0)
        o2.step()
        assert torch.isfinite(n) and lin.weight.grad is not None
        # (c) SGD without nesterov on the model: torch's step
        m2 = _model()
        o3 = torch.optim.SGD(m2.parameters(), 1e-2, momentum=0.9)
        _step(m2, o3, 4, 20.0)
        assert any(p.grad is not None for p in m2.parameters())
    finally:
        accel.uninstall()
