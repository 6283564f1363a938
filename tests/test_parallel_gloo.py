"""N > 1 path on CPU: two gloo ranks, uneven shards.  Each rank differentiates its shard
of the batch with losses divided by the GLOBAL row counts (ta3n_amd.parallel.
loss_normalisers - the numbers TrainEngine.set_hyper feeds the HIP loss kernel), packs the
gradients into the plan's flat live buffer and the product's single SUM all-reduce must
reproduce the single-process global-batch gradient of the oracle (what the reference's
DataParallel gather + global means compute, main.py:446, 533; loss.py:24)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ta3n_oracle as orc
from ta3n_amd import _lib, parallel
from ta3n_amd.synthetic import synth_batch, synth_state

CFG = dict(C=7, T=4, D=512, fc=32, Bs=5, Bt=3)
FLAGS = (_lib.FLAG_ADV_RELATION | _lib.FLAG_ADV_VIDEO | _lib.FLAG_ADV_FRAME | _lib.FLAG_ATTN_ENTROPY | _lib.FLAG_TRANS_ATTN)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _flat_grads(plan, grads):
    flat = torch.zeros(plan.live_floats, dtype=torch.float64)
    for name, off, shape, live in plan.params:
        if live and name in grads:
            n = int(np.prod(shape))
            flat[off:off + n] = grads[name].reshape(-1)
    return flat


def _shard_grad(params, cfg, xs, xt, ys, beta, gamma, norm, T):
    """sum over local rows / GLOBAL counts, written with the oracle's mean-reduced parts."""
    p = {k: v.detach().clone().double().requires_grad_(True) for k, v in params.items()}
    ns, nt = xs.size(0), xt.size(0)
    if ns + nt == 0:
        return {}
    src = orc.forward_domain(p, xs.double(), beta, cfg) if ns else None
    tgt = orc.forward_domain(p, xt.double(), beta, cfg) if nt else None
    loss = 0
    if ns:
        loss = loss + torch.nn.functional.cross_entropy(src["out"], ys, reduction="sum") * norm["inv_n_cls"]
    for l, key in enumerate(("inv_n_rel", "inv_n_vid", "inv_n_frm")):
        for dom, lab in ((src, 0), (tgt, 1)):
            if dom is not None:
                z = dom["pred_domain"][l].reshape(-1, 2)
                loss = loss + torch.nn.functional.cross_entropy(z, torch.full((z.size(0),), lab), reduction="sum") * norm[key]
    for dom in (src, tgt):
        if dom is not None:
            n = dom["out"].size(0)
            loss = loss + gamma * orc.attentive_entropy(dom["out"], dom["pred_domain"][1]) * n * norm["inv_n_ent"]
    names = [k for k in p if orc.is_live(k)]
    g = torch.autograd.grad(loss, [p[k] for k in names], allow_unused=True)
    return {k: gi for k, gi in zip(names, g) if gi is not None}


def _worker(rank, world, port, out, c=None):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    r, lr_, w = parallel.init_distributed("gloo")
    assert (r, w) == (rank, world)
    c = c or CFG
    cfg = orc.Config(num_class=c["C"], num_segments=c["T"], feature_dim=c["D"], fc_dim=c["fc"], dropout_i=0, dropout_v=0)
    params = synth_state(orc.param_shapes(cfg), seed=5)
    xs, xt, ys, yt = synth_batch(c["C"], c["T"], c["D"], c["Bs"], c["Bt"], seed=9)
    plan = _lib.Plan(parallel.padded_shard_size(c["Bs"], world), parallel.padded_shard_size(c["Bt"], world), c["T"],
                     c["D"], c["fc"], c["C"], FLAGS)
    lo, hi = parallel.shard_range(c["Bs"], world, rank)
    lo_t, hi_t = parallel.shard_range(c["Bt"], world, rank)
    gs, gt = parallel.global_counts(hi - lo, hi_t - lo_t)
    assert (gs, gt) == (c["Bs"], c["Bt"])
    norm = parallel.loss_normalisers(gs, gt, c["T"])
    g = _shard_grad(params, cfg, xs[lo:hi], xt[lo_t:hi_t], ys[lo:hi], [0.75, 0.75, 0.5], 0.3, norm, c["T"])
    flat = _flat_grads(plan, g)
    parallel.all_reduce_sum_(flat)
    if rank == 0:
        torch.save(flat, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sum_allreduce_equals_global_batch_gradient(tmp_path):
    out = str(tmp_path / "flat.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    flat = torch.load(out)
    c = CFG
    cfg = orc.Config(num_class=c["C"], num_segments=c["T"], feature_dim=c["D"], fc_dim=c["fc"], dropout_i=0, dropout_v=0)
    params = {k: v.double() for k, v in synth_state(orc.param_shapes(cfg), seed=5).items()}
    xs, xt, ys, yt = synth_batch(c["C"], c["T"], c["D"], c["Bs"], c["Bt"], seed=9)
    state = orc.TrainState(params=params, lr=0.0)
    res = orc.train_step(state, xs.double(), xt.double(), ys, [0.75, 0.75, 0.5], 0.3, cfg, clip=None)
    plan = _lib.Plan(3, 2, c["T"], c["D"], c["fc"], c["C"], FLAGS)
    ref = _flat_grads(plan, res["grads"])
    assert ref.abs().max() > 0
    assert torch.allclose(flat, ref, rtol=1e-9, atol=1e-12), (flat - ref).abs().max()


CFG4 = dict(C=7, T=3, D=128, fc=16, Bs=10, Bt=74)


def test_four_ranks_with_the_headline_target_count_equal_the_global_batch_gradient(tmp_path):
    """74 target videos over 4 ranks -> shards of 19 / 19 / 18 / 18 inside a static 19-row buffer (zero-padded dummy rows, the
    reference's rule main.py:366-372), 10 source videos -> 3 / 3 / 2 / 2: per-rank sum-losses over the GLOBAL counts + ONE sum
    all-reduce reproduce the single-process global-batch gradient."""
    c = CFG4
    spans = [parallel.shard_range(c["Bt"], 4, r) for r in range(4)]
    assert [e - b for b, e in spans] == [19, 19, 18, 18] and parallel.padded_shard_size(c["Bt"], 4) == 19
    assert [e - b for b, e in (parallel.shard_range(c["Bs"], 4, r) for r in range(4))] == [3, 3, 2, 2]
    out = str(tmp_path / "flat4.pt")
    mp.spawn(_worker, args=(4, _free_port(), out, c), nprocs=4, join=True)
    flat = torch.load(out)
    cfg = orc.Config(num_class=c["C"], num_segments=c["T"], feature_dim=c["D"], fc_dim=c["fc"], dropout_i=0, dropout_v=0)
    params = {k: v.double() for k, v in synth_state(orc.param_shapes(cfg), seed=5).items()}
    xs, xt, ys, yt = synth_batch(c["C"], c["T"], c["D"], c["Bs"], c["Bt"], seed=9)
    res = orc.train_step(orc.TrainState(params=params, lr=0.0), xs.double(), xt.double(), ys, [0.75, 0.75, 0.5], 0.3, cfg, clip=None)
    plan = _lib.Plan(3, 19, c["T"], c["D"], c["fc"], c["C"], FLAGS)
    ref = _flat_grads(plan, res["grads"])
    assert ref.abs().max() > 0
    assert torch.allclose(flat, ref, rtol=1e-9, atol=1e-12), (flat - ref).abs().max()


def test_shard_ranges_cover_batch_without_overlap():
    for n in (0, 1, 5, 74, 128, 202):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(e - b for b, e in spans) <= parallel.padded_shard_size(n, world)
            assert max(e - b for b, e in spans) - min(e - b for b, e in spans) <= 1
    # 74 target videos over 8 ranks: 10 per rank static, two ranks carry one dummy row less
    assert parallel.padded_shard_size(74, 8) == 10
    n = parallel.loss_normalisers(128, 74, 5)
    assert n["inv_n_cls"] == 1 / 128 and n["inv_n_rel"] == 1 / (202 * 4) and n["inv_n_frm"] == 1 / (202 * 5)


def _comm_agreement_worker(rank, world, port, fail_rank, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L = _lib.lib()

    class Fake:      # the two C entry points NativeComm uses, with a failure injected on ONE rank
        def __init__(self, real):
            self._real = real

        def __getattr__(self, k):
            return getattr(self._real, k)

        def ta3n_comm_unique_id(self, buf):
            return -2 if (fail_rank == ("id", rank)) else 0

        def ta3n_comm_create(self, *a):
            return -2 if (fail_rank == ("create", rank)) else 0

        def ta3n_comm_destroy(self, h):
            out.put(("destroyed", rank))

        def ta3n_last_error(self):
            return b"injected failure"
    fake = Fake(L)
    _lib._LIB = fake
    try:
        try:
            parallel.NativeComm(None, None)
            out.put(("ok", rank))
        except RuntimeError as ex:
            out.put(("raised", rank, str(ex)))
    finally:
        _lib._LIB = L
        dist.destroy_process_group()


@pytest.mark.parametrize("where", ["id", "create"])
def test_native_comm_failure_on_one_rank_is_raised_on_every_rank(where):
    """if the library's RCCL communicator cannot be created on SOME rank, every rank must learn it and fall
    back together - otherwise the ranks enqueue different collectives and hang."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_comm_agreement_worker, args=(r, 2, port, (where, 1), q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = []
    while not q.empty():
        got.append(q.get())
    raised = sorted(g[1] for g in got if g[0] == "raised")
    assert raised == [0, 1], got
    assert all("rank(s) [1]" in g[2] for g in got if g[0] == "raised"), got


# ---- discrepancy loss (dis_DA DAN / JAN) over ranks: the global-batch loss of the reference's DataParallel gather ----
DIS = dict(C=7, Fv=24, Bs=6, Bt=5)      # per-rank padded batch; valid rows per rank below (uneven on purpose)
DIS_VALID = [(6, 5), (4, 5)]             # (source, target) real rows on rank 0 / rank 1: 10 + 10 global, size = 10


def _dis_inputs():
    g = torch.Generator().manual_seed(123)
    c = DIS
    ys = [torch.randn(c["Bs"] + c["Bt"], c["C"], generator=g, dtype=torch.float64) for _ in range(2)]
    vs = [torch.randn(c["Bs"] + c["Bt"], c["Fv"], generator=g, dtype=torch.float64) for _ in range(2)]
    return ys, vs


def _dis_worker(rank, world, port, dis_DA, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    ys, vs = _dis_inputs()
    ns, nt = DIS_VALID[rank]
    loss, gy, gv = parallel.discrepancy_over_ranks(dis_DA, ("Y", "Y", "N"), 0.7, ys[rank], vs[rank], DIS["Bs"], ns, nt)
    torch.save((loss, gy, gv), f"{out}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("dis_DA", ["DAN", "JAN"])
def test_two_rank_discrepancy_is_the_global_batch_loss_and_each_rank_keeps_its_own_gradient_rows(tmp_path, dis_DA):
    """Reference: main.py:452-505 on the outputs nn.DataParallel gathered in replica order.  Single-process restatement here: concatenate
    the ranks' valid rows, take the loss on the first min(#source, #target) videos, differentiate."""
    from ta3n_amd import loss as L
    out = str(tmp_path / "dis.pt")
    mp.spawn(_dis_worker, args=(2, _free_port(), dis_DA, out), nprocs=2, join=True)
    ys, vs = _dis_inputs()
    Bs, C = DIS["Bs"], DIS["C"]
    src_y = torch.cat([ys[r][:DIS_VALID[r][0]] for r in range(2)]).requires_grad_(True)
    src_v = torch.cat([vs[r][:DIS_VALID[r][0]] for r in range(2)]).requires_grad_(True)
    tgt_y = torch.cat([ys[r][Bs:Bs + DIS_VALID[r][1]] for r in range(2)]).requires_grad_(True)
    tgt_v = torch.cat([vs[r][Bs:Bs + DIS_VALID[r][1]] for r in range(2)]).requires_grad_(True)
    size = min(src_y.size(0), tgt_y.size(0))
    fs, ft = [src_y[:size], src_v[:size]], [tgt_y[:size], tgt_v[:size]]
    if dis_DA == "JAN":
        ref = L.JAN(fs, ft, kernel_muls=[2.0, 2.0], kernel_nums=[2, 5], fix_sigma_list=[None, None], ver=2)
    else:
        ref = sum(L.mmd_rbf(fs[l], ft[l], kernel_mul=2.0, kernel_num=[2, 5][l], fix_sigma=None, ver=2) for l in range(2))
    g = torch.autograd.grad(0.7 * ref, (src_y, src_v, tgt_y, tgt_v))
    s0 = t0 = 0
    for r in range(2):
        loss, gy, gv = torch.load(f"{out}.{r}")
        ns, nt = DIS_VALID[r]
        assert torch.allclose(loss, ref.detach(), rtol=1e-12, atol=1e-14)
        assert torch.allclose(gy[:ns], g[0][s0:s0 + ns], rtol=1e-10, atol=1e-14) and torch.allclose(gv[:ns], g[1][s0:s0 + ns], rtol=1e-10, atol=1e-14)
        assert torch.allclose(gy[Bs:Bs + nt], g[2][t0:t0 + nt], rtol=1e-10, atol=1e-14) and torch.allclose(gv[Bs:Bs + nt], g[3][t0:t0 + nt], rtol=1e-10, atol=1e-14)
        assert gy[ns:Bs].abs().sum() == 0 and gy[Bs + nt:].abs().sum() == 0 and gv[ns:Bs].abs().sum() == 0      # padding rows take no part
        assert g[0].abs().max() > 0
        s0, t0 = s0 + ns, t0 + nt


def test_one_rank_discrepancy_matches_the_plain_form():
    """No process group: the same function is the single-rank path of TrainEngine.discrepancy()."""
    from ta3n_amd import loss as L
    ys, vs = _dis_inputs()
    Bs = DIS["Bs"]
    loss, gy, gv = parallel.discrepancy_over_ranks("DAN", ("N", "Y", "N"), 0.5, ys[0], vs[0], Bs, 6, 5)
    v = vs[0].clone().requires_grad_(True)
    ref = L.mmd_rbf(v[:5], v[Bs:Bs + 5], kernel_mul=2.0, kernel_num=5, fix_sigma=None, ver=2)
    gref, = torch.autograd.grad(0.5 * ref, v)
    assert torch.allclose(loss, ref.detach()) and torch.allclose(gv, gref) and gy.abs().max() == 0
