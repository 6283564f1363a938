"""The DA options of TrainEngine under more than one rank, checked on ONE GPU by running the ranks one after the other: two engines take
the two halves of a batch with the job-wide counts in their loss normalisers (set_hyper(global_source=, global_target=) - what
train_ddp.py passes), and the SUM of their gradients - what the step's all-reduce produces - must be the gradient of one engine on the
whole batch.  That is the reference's semantics: nn.DataParallel scatters the batch, gathers the outputs and takes every loss as a
mean over the gathered global batch (main.py:79, 446-562).
  * ens_DA MCD: the classifier discrepancy is a mean over the GLOBAL target batch (loss.py:29-30).
  * dis_DA: the loss couples all videos; its rank-side is parallel.discrepancy_over_ranks (tests/test_parallel_gloo.py, two gloo
    ranks); here the engine's single-rank path runs through the same function.
  * use_bn: statistics are per replica under DataParallel, i.e. per rank - nothing to sum; sync_buffers() is the broadcast of
    replica 0's running statistics (a no-op on one rank)."""
import pytest
import torch

from ta3n_amd import _lib
from ta3n_amd.engine import ALL_FLAGS, TrainEngine
from ta3n_amd.synthetic import synth_batch, synth_state

pytestmark = pytest.mark.gpu

C, T, D, F = 7, 5, 256, 64
BS, BT = 8, 6


def _engine(bs, bt, flags, **kw):
    eng = TrainEngine(bs, bt, T, D, F, C, flags=flags, dropout_i=0.0, dropout_v=0.0, clip=None, **kw)
    shapes = {n: s for n, _, s, _ in eng.plan.params}
    eng.load_state(synth_state(shapes, seed=3, scale="trained"))
    return eng


def _grads(eng, xs, xt, ys, **hyper_kw):
    eng.set_batch(xs.cuda(), xt.cuda(), ys.cuda())
    eng.train_step([0.75, 0.75, 0.5], 0.3, 0.0, **hyper_kw)
    torch.cuda.synchronize()
    live = set(eng.live_names())
    return {k: v.detach().double().clone() for k, v in eng.param_views(eng.G).items() if k in live}


@pytest.mark.parametrize("entropy", [True, False], ids=["with_attentive_entropy", "without"])
def test_mcd_gradients_of_two_half_batches_sum_to_the_whole_batch_gradient(entropy):
    flags = ALL_FLAGS if entropy else ALL_FLAGS & ~_lib.FLAG_ATTN_ENTROPY
    xs, xt, ys, _ = synth_batch(C, T, D, BS, BT, seed=17)
    whole = _engine(BS, BT, flags, ens_DA="MCD", mu=0.7)
    g_all = _grads(whole, xs, xt, ys)
    loss_s_all, loss_c2_all = whole.loss_s.item(), whole.loss_c2.item()
    hs, ht = BS // 2, BT // 2
    total, loss_s, loss_c2 = None, 0.0, 0.0
    for r in range(2):
        half = _engine(hs, ht, flags, ens_DA="MCD", mu=0.7)
        g = _grads(half, xs[r * hs:(r + 1) * hs], xt[r * ht:(r + 1) * ht], ys[r * hs:(r + 1) * hs], global_source=BS, global_target=BT)
        total = g if total is None else {k: total[k] + g[k] for k in g}
        loss_s += half.loss_s.item(); loss_c2 += half.loss_c2.item()      # partial sums over the rank's rows / the global count
    assert abs(loss_s - loss_s_all) < 1e-6 and abs(loss_c2 - loss_c2_all) < 1e-5, (loss_s, loss_s_all, loss_c2, loss_c2_all)
    assert set(total) == set(g_all)
    for k in g_all:
        scale = max(g_all[k].abs().max().item(), 1e-12)
        assert (total[k] - g_all[k]).abs().max().item() <= 2e-5 * scale + 1e-9, (k, scale, (total[k] - g_all[k]).abs().max().item())


@pytest.mark.parametrize("dis_DA", ["DAN", "JAN"])
def test_engine_discrepancy_runs_through_the_rank_aware_function(dis_DA):
    """One rank: TrainEngine.discrepancy() == parallel.discrepancy_over_ranks without a group (the DAN / JAN goldens of the reference
    gate the same path in tests/test_gpu_da_extras.py); the loss and its logit / feature gradients are finite and non-trivial."""
    xs, xt, ys, _ = synth_batch(C, T, D, BS, BT, seed=19)
    eng = _engine(BS, BT, ALL_FLAGS, dis_DA=dis_DA, place_dis=("Y", "Y", "N"), alpha=0.5)
    _grads(eng, xs, xt, ys)
    assert eng.loss_d is not None and torch.isfinite(eng.loss_d) and eng.loss_d.item() != 0.0
    assert eng.region("gV_ext", (eng.B, -1)).abs().max().item() > 0


def test_sync_buffers_is_a_no_op_on_one_rank():
    eng = _engine(BS, BT, ALL_FLAGS, use_bn="AdaBN")
    xs, xt, ys, _ = synth_batch(C, T, D, BS, BT, seed=23)
    _grads(eng, xs, xt, ys)
    before, n = eng.bn_running.clone(), eng.bn_batches
    eng.sync_buffers()
    assert torch.equal(before, eng.bn_running) and eng.bn_batches == n and n == 1
