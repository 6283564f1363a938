"""Chained launches on the GPU (ta3n_config.chain): the same tiles with the same arithmetic as one launch per level, handed
over inside the launch - so with one tile shape for every launch the results must be BIT-identical to the unchained plan, in
every arithmetic, over several steps, eager and pipelined; hand-off bookkeeping must come back clean (ta3n_chain_status)."""
import pytest
import torch

from golden_util import Golden, case_config
from ta3n_amd.engine import TrainEngine
from ta3n_amd.synthetic import synth_batch, synth_state

pytestmark = pytest.mark.gpu_ab      # measured-and-rejected variant / opt-in transport: `pytest -m gpu_ab` on the experiments build (tests/conftest.py)

ARITH = {"f32": {}, "bf16": dict(bf16=True, bf16_store=True), "bf16_cvt": dict(bf16=True), "f32x3": dict(f32_split=True)}


def _run(shape, arith, chain, tile, mode, steps=4, dropout=0.5):
    Bs, Bt, T, D, F, C = shape
    eng = TrainEngine(Bs, Bt, T, D, F, C, dropout_i=dropout, dropout_v=dropout, tile_config=tile, chain=chain, **ARITH[arith])
    shapes = {n: s for n, _, s, _ in eng.plan.params}
    eng.load_state(synth_state(shapes, seed=7))
    for i in range(steps):
        xs, xt, ys, yt = synth_batch(C, T, D, Bs, Bt, seed=5 + i)
        eng.set_batch(xs.cuda(), xt.cuda(), ys.cuda())
        args = ([0.75, 0.75, 0.5], 0.003, 1e-3 * (i + 1))
        if mode == "pipelined":
            eng.train_step_pipelined(*args, seed=i)
        else:
            eng.train_step(*args, seed=i)
    eng.flush()
    torch.cuda.synchronize()
    eng.chain_status()
    return eng.P.clone(), eng.M.clone(), eng.G.clone(), eng.region("losses")[:6].clone(), eng.outputs()["out"].clone()


@pytest.mark.parametrize("mode", ["plain", "pipelined"])
@pytest.mark.parametrize("arith", sorted(ARITH))
@pytest.mark.parametrize("shape,tile", [((6, 4, 5, 512, 64, 12), 124), ((40, 30, 3, 256, 128, 7), 222), ((33, 37, 9, 192, 64, 30), 114),
                                        ((128, 74, 5, 2048, 512, 12), 124)])
def test_chained_step_is_bit_identical_to_one_launch_per_level(shape, tile, arith, mode):
    a = _run(shape, arith, False, tile, mode)
    b = _run(shape, arith, True, tile, mode)
    for x, y, what in zip(a, b, ("params", "momentum", "grads", "losses", "logits")):
        assert torch.equal(x, y), what
    assert torch.isfinite(a[0]).all()


@pytest.mark.parametrize("arith", ["bf16", "f32"])
def test_chained_step_repeated_many_times_keeps_its_counters_clean(arith):
    """200 pipelined steps at the headline shape with the tuned tiles, all enqueued at once (the GPU runs them back to back):
    hand-offs clean and BIT-identical to the unchained plan given the tile shapes the chained launches use (a chained launch has
    one shape for all its levels: the first level's)."""
    from ta3n_amd.engine import lr_dann
    from ta3n_amd.tuning import tuned_phase_tiles
    bf16 = arith == "bf16"
    tiles = tuned_phase_tiles(202, 5, 2048, 512, bf16, bf16)
    same = list(tiles)
    same[11] = same[12] = same[10]
    same[15] = same[14]
    outs = []
    for chain, pt in ((False, same), (True, tiles)):
        eng = TrainEngine(128, 74, 5, 2048, 512, 12, chain=chain, phase_tiles=pt, **ARITH[arith])
        shapes = {n: s for n, _, s, _ in eng.plan.params}
        eng.load_state(synth_state(shapes, seed=7))
        xs, xt, ys, yt = synth_batch(12, 5, 2048, 128, 74, seed=1)
        eng.set_batch(xs.cuda(), xt.cuda(), ys.cuda())
        eng.train_steps([([0.75, 0.75, 0.5], 0.003, lr_dann(3e-2, i / 360.0)) for i in range(200)])
        eng.flush()
        torch.cuda.synchronize()
        eng.chain_status()
        assert torch.isfinite(eng.P).all()
        outs.append((eng.P.clone(), eng.M.clone(), eng.region("losses")[:6].clone()))
    for x, y in zip(*outs):
        assert torch.equal(x, y)
