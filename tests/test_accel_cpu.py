"""ta3n_amd.accel on CPU tensors (no HIP call involved): VideoModel's parameters as views of one flat buffer, gradients handed out by
models._deliver_grads as views of another, then torch.nn.utils.clip_grad_norm_ + torch.optim.SGD.step through the flat passes
against torch's own per-tensor code on a copy - same numbers, optimiser state_dict interchangeable, torch's code when a precondition
fails."""
import copy

import torch

from ta3n_amd import accel
from ta3n_amd.models import VideoModel, _backward_buffer, _deliver_grads

CPU = torch.device("cpu")


def _model():
    torch.manual_seed(0)
    m = VideoModel(12, "video", "trn-m", "RGB", train_segments=5, val_segments=5, base_model="resnet18", fc_dim=64, verbose=False)
    plan = m._plan(6, 4)
    m._ensure_flat(plan, CPU)
    return m, plan


def _fake_backward(m, plan, seed, unused=()):
    g, fresh = _backward_buffer(m, plan, CPU)
    gen = torch.Generator().manual_seed(seed)
    for name, off, shape, live in plan.params:
        n = 1
        for s in shape:
            n *= s
        if live:
            g[off:off + n] = torch.randn(n, generator=gen)
    _deliver_grads(m, plan, g, fresh, tuple(unused), [True] * len(plan.params))


def test_flat_clip_and_step_equal_torchs_per_tensor_code():
    fast, plan = _model()
    ref, plan_r = _model()                 # the same initialisation (seeded)
    assert all(torch.equal(a, b) for a, b in zip(fast.parameters(), ref.parameters()))
    o_fast = torch.optim.SGD(fast.parameters(), 3e-2, momentum=0.9, weight_decay=1e-4, nesterov=True)
    o_ref = torch.optim.SGD(ref.parameters(), 3e-2, momentum=0.9, weight_decay=1e-4, nesterov=True)
    try:
        for step, max_norm in enumerate((1e9, 1e9, 5.0, 5.0)):
            accel.uninstall()
            o_ref.zero_grad()
            _fake_backward(ref, plan_r, 10 + step)
            n_ref = torch.nn.utils.clip_grad_norm_(ref.parameters(), max_norm)
            o_ref.step()
            assert accel.install()
            o_fast.zero_grad()
            _fake_backward(fast, plan, 10 + step)
            views_before = [p.grad for p in fast.parameters() if p.grad is not None]
            n_fast = torch.nn.utils.clip_grad_norm_(fast.parameters(), max_norm)
            o_fast.step()
            assert fast._mom_flat is not None                                     # the flat paths ran
            assert [p.grad for p in fast.parameters() if p.grad is not None] == views_before or \
                all(a is b for a, b in zip([p.grad for p in fast.parameters() if p.grad is not None], views_before))
            assert abs(float(n_ref) - float(n_fast)) <= 1e-5 * float(n_ref)
            for (k, a), (_, b) in zip(ref.named_parameters(), fast.named_parameters()):
                if max_norm > 1e6 and step < 2:
                    assert torch.equal(a, b), k                                   # no clipping so far: the same operations element for element
                else:
                    assert torch.allclose(a, b, rtol=1e-5, atol=1e-7), k
        # one persistent gradient buffer: the views of the last backward are the views of the first
        assert fast._grad_flat is fast._grad_buf
        # optimiser state: same keys, same values; it loads into a fresh torch optimiser
        sd_f, sd_r = o_fast.state_dict(), o_ref.state_dict()
        assert sd_f["state"].keys() == sd_r["state"].keys()
        for i in sd_r["state"]:
            assert torch.allclose(sd_f["state"][i]["momentum_buffer"], sd_r["state"][i]["momentum_buffer"], rtol=1e-5, atol=1e-5)
        o_new = torch.optim.SGD(ref.parameters(), 1e-2, momentum=0.9, nesterov=True)
        o_new.load_state_dict(copy.deepcopy(sd_f))
    finally:
        accel.uninstall()


def test_second_backward_before_zero_grad_accumulates_in_place():
    m, plan = _model()
    _fake_backward(m, plan, 1)
    first = m._grad_flat[: plan.live_floats].clone()
    g_obj = next(p.grad for p in m.parameters() if p.grad is not None)
    _fake_backward(m, plan, 2, unused=("fc_feature_domain.",))                    # MCD's second pass: some logits without a loss
    second = torch.zeros_like(first)
    gen = torch.Generator().manual_seed(2)
    for name, off, shape, live in plan.params:
        n = 1
        for s in shape:
            n *= s
        if live:
            second[off:off + n] = torch.randn(n, generator=gen)
    assert torch.allclose(m._grad_flat[: plan.live_floats], first + second)
    assert next(p.grad for p in m.parameters() if p.grad is not None) is g_obj    # the same view objects


def test_torchs_code_runs_when_a_live_parameter_has_no_gradient_or_the_optimiser_is_another():
    m, plan = _model()
    try:
        assert accel.install()
        opt = torch.optim.SGD(m.parameters(), 3e-2, momentum=0.9, weight_decay=1e-4, nesterov=True)
        before = {k: v.detach().clone() for k, v in m.named_parameters()}
        _fake_backward(m, plan, 3, unused=("fc_feature_domain.", "fc_classifier_domain."))
        torch.nn.utils.clip_grad_norm_(m.parameters(), 20.0)
        opt.step()
        after = dict(m.named_parameters())
        assert m._mom_flat is None                                                # torch's own step ran
        assert torch.equal(before["fc_feature_domain.weight"], after["fc_feature_domain.weight"])          # untouched: no weight decay either
        assert not torch.equal(before["fc_feature_shared_source.weight"], after["fc_feature_shared_source.weight"])
        opt.zero_grad()
        opt2 = torch.optim.SGD(m.parameters(), 1e-2, momentum=0.9)                  # no nesterov: torch's step
        _fake_backward(m, plan, 4)
        opt2.step()
        assert m._mom_flat is None
        lin = torch.nn.Linear(4, 3)
        lin(torch.randn(2, 4)).sum().backward()
        assert torch.isfinite(torch.nn.utils.clip_grad_norm_(lin.parameters(), 1.0))
    finally:
        accel.uninstall()
