"""ta3n_config.split_k = 2 (every tile of the gradient at the frame features computed by two workgroups over halves of its K
segments; partial tile + ticket, the second to arrive finishes): same numbers as the unsplit step up to fp32 summation order,
bit-identical from run to run (a + b = b + a: it does not matter which half arrives last), tickets back at zero."""
import pytest
import torch

from golden_util import Golden, case_config
from ta3n_amd.engine import TrainEngine
from ta3n_amd.synthetic import synth_batch, synth_state

pytestmark = pytest.mark.gpu_ab      # measured-and-rejected variant / opt-in transport: `pytest -m gpu_ab` on the experiments build (tests/conftest.py)

ARITH = {"f32": {}, "bf16": dict(bf16=True, bf16_store=True), "f32x3p": dict(f32_split=True, bf16_store=True)}


@pytest.mark.parametrize("mode", [2, 4])      # 2: the gradient at the frame features; 4 (round 6): the shared-FC product's single K segment halved
@pytest.mark.parametrize("arith", list(ARITH))
@pytest.mark.parametrize("name", ["tiny_T5", "tiny_T9", "headline"])
def test_split_k_matches_the_unsplit_step_and_is_reproducible(name, arith, mode):
    g = Golden(name)
    c = case_config(g)
    T = c["T"]
    res = {}
    for split in (0, mode, mode):
        eng = TrainEngine(c["Bs"], c["Bt"], T, c["D"], c["fc_dim"], c["C"], dropout_i=0.5, dropout_v=0.5, clip=c["clip"], split_k=split,
                          **ARITH[arith])
        assert ("splitk_part" in eng.plan.regions) == (split != 0)
        shapes = {n: s for n, _, s, _ in eng.plan.params}
        eng.load_state(synth_state(shapes, seed=c["wseed"], scale=c["wscale"]))
        for step in range(4):
            xs, xt, ys, yt = synth_batch(c["C"], T, c["D"], c["Bs"], c["Bt"], seed=100 + step)
            eng.set_batch(xs.cuda(), xt.cuda(), ys.cuda())
            eng.train_step_pipelined([0.75, 0.75, 0.5], 0.003, 0.01)
        eng.flush()
        torch.cuda.synchronize()
        if split:
            assert eng.region("splitk_ticket").view(torch.int32).abs().max().item() == 0
        res.setdefault(split, []).append((eng.P.clone(), eng.G.clone(), eng.region("losses")[:6].clone()))
    (p0, g0, l0), = res[0]
    (p1, g1, l1), (p2, g2, l2) = res[mode]
    assert torch.equal(p1, p2) and torch.equal(g1, g2) and torch.equal(l1, l2)          # run to run
    scale = p0.abs().max().item()
    # stored bf16 planes: a summation-order ulp can flip a bf16 rounding (of hi, or of lo) or a ReLU, which four updates amplify
    # (see test_gpu_bf16.py::test_twin_storage_is_the_same_arithmetic_over_several_updates)
    tol = 2e-3 if arith == "bf16" else 1e-3 if arith == "f32x3p" else 1e-4
    assert (p0 - p1).abs().max().item() <= tol * scale and (p0 - p1).abs().mean().item() <= 0.02 * tol * scale
    # (mode 4 in bf16: the split changes the summation order of the FIRST layer, whose bf16 twin feeds everything behind it - a rounding flip
    # there moves the loss scalars of the fourth step by up to ~1 %)
    assert torch.allclose(l0, l1, rtol=(2e-2 if mode == 4 else 5e-3) if arith == "bf16" else 1e-4, atol=1e-4)
