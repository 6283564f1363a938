"""GPU: the N > 1 code path of TrainEngine on ONE GPU - two ranks (gloo over CUDA tensors; RCCL refuses two
ranks on one device) each step their shard with global loss normalisers, sum-all-reduce the flat gradient
buffer and update; both must end on the parameters of a single process that stepped the whole batch
(what the reference's DataParallel gather + global means compute)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ta3n_amd.synthetic import synth_batch, synth_state

pytestmark = pytest.mark.gpu

CFG = dict(C=12, T=5, D=512, fc=128, Bs=12, Bt=8)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _engine(Bs, Bt, mode):
    """mode: 'fused' / 'unfused' (fp32) or 'bf16' (bf16 MFMA operands from twins, update pipelined into the next step)."""
    from ta3n_amd.engine import TrainEngine
    c = CFG
    return TrainEngine(Bs, Bt, c["T"], c["D"], c["fc"], c["C"], dropout_i=0.0, dropout_v=0.0, fused=(mode != "unfused"),
                       bf16=mode.endswith("bf16"), bf16_store=mode.endswith("bf16"))


def _run(eng, xs, xt, ys, steps, mode, **kw):
    for i in range(steps):
        eng.set_batch(xs.cuda(), xt.cuda(), ys.cuda())
        if mode.endswith("bf16"):
            eng.train_step_pipelined([0.75, 0.75, 0.5], 0.003, 1e-2, seed=i, **kw)
        else:
            eng.train_step([0.75, 0.75, 0.5], 0.003, 1e-2, seed=i, **kw)
    eng.flush()
    torch.cuda.synchronize()
    return eng.P.detach().cpu().clone()


def _worker(rank, world, port, out, mode):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      TA3N_DDP_BUCKETS="2" if mode == "fused" else "1",      # cover both gradient bucketings
                      TA3N_DDP_SHARDED="1" if mode.startswith("sharded") else "0")      # ... and the sharded update (own-shard optimiser)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ta3n_amd import parallel
    c = CFG
    xs, xt, ys, yt = synth_batch(c["C"], c["T"], c["D"], c["Bs"], c["Bt"], seed=9)
    lo, hi = parallel.shard_range(c["Bs"], world, rank)
    lo_t, hi_t = parallel.shard_range(c["Bt"], world, rank)
    eng = _engine(hi - lo, hi_t - lo_t, mode)
    assert eng.world == world and eng._sharded == mode.startswith("sharded")
    shapes = {n: s for n, _, s, _ in eng.plan.params}
    eng.load_state(synth_state(shapes, seed=3))
    P = _run(eng, xs[lo:hi], xt[lo_t:hi_t], ys[lo:hi], 3, mode, global_source=c["Bs"], global_target=c["Bt"])
    torch.save(P, f"{out}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["fused", "unfused", "bf16", "sharded", "sharded_bf16"])      # (sharded update: opt-in, shipped in the default library)
def test_two_ranks_equal_single_process_global_batch(tmp_path, mode):
    c = CFG
    out = str(tmp_path / "P")
    mp.spawn(_worker, args=(2, _free_port(), out, mode), nprocs=2, join=True)
    p0, p1 = torch.load(out + ".0"), torch.load(out + ".1")
    xs, xt, ys, yt = synth_batch(c["C"], c["T"], c["D"], c["Bs"], c["Bt"], seed=9)
    eng = _engine(c["Bs"], c["Bt"], mode)
    shapes = {n: s for n, _, s, _ in eng.plan.params}
    eng.load_state(synth_state(shapes, seed=3))
    ref = _run(eng, xs, xt, ys, 3, mode)
    assert torch.equal(p0, p1)                                    # every rank applies the identical update
    # bf16: the sharded weight gradients sum the same bf16 products in another order; a sum-order ulp can flip a bf16 rounding
    rtol, atol = (2e-4, 2e-6) if not mode.endswith("bf16") else (5e-3, 5e-5)
    assert torch.allclose(p0, ref, rtol=rtol, atol=atol), (p0 - ref).abs().max()
    moved = (ref - synth_flat(eng, shapes)).abs().max()
    assert moved > 1e-4                                           # the steps actually changed the parameters


def synth_flat(eng, shapes):
    from ta3n_amd.engine import TrainEngine
    e2 = TrainEngine(eng.Bs, eng.Bt, eng.T, eng.D, CFG["fc"], eng.C, dropout_i=0.0, dropout_v=0.0)
    e2.load_state(synth_state(shapes, seed=3))
    return e2.P.detach().cpu().clone()


# ---- the DA options under two ranks (round 4): nn.DataParallel's semantics ----
DA_OPTS = {
    "mcd": dict(ens_DA="MCD", mu=0.7),
    "dan": dict(dis_DA="DAN", place_dis=("Y", "Y", "N"), alpha=0.5),
    "jan_mcd": dict(dis_DA="JAN", place_dis=("Y", "Y", "N"), alpha=0.5, ens_DA="MCD", mu=0.7),
    "adabn": dict(use_bn="AdaBN"),
}


def _engine_da(Bs, Bt, opt):
    from ta3n_amd.engine import TrainEngine
    c = CFG
    return TrainEngine(Bs, Bt, c["T"], c["D"], c["fc"], c["C"], dropout_i=0.0, dropout_v=0.0, **DA_OPTS[opt])


def _run_da(eng, xs, xt, ys, steps, **kw):
    for i in range(steps):
        eng.set_batch(xs.cuda(), xt.cuda(), ys.cuda())
        eng.train_step([0.75, 0.75, 0.5], 0.003, 1e-2, seed=i, **kw)
    eng.flush()
    eng.sync_buffers()
    torch.cuda.synchronize()
    extra = eng.bn_running.detach().cpu().clone() if eng.bn_running is not None else torch.zeros(0)
    return eng.P.detach().cpu().clone(), extra, float(eng.loss_d) if eng.loss_d is not None else 0.0


def _worker_da(rank, world, port, out, opt):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ta3n_amd import parallel
    c = CFG
    xs, xt, ys, yt = synth_batch(c["C"], c["T"], c["D"], c["Bs"], c["Bt"], seed=9)
    lo, hi = parallel.shard_range(c["Bs"], world, rank)
    lo_t, hi_t = parallel.shard_range(c["Bt"], world, rank)
    eng = _engine_da(hi - lo, hi_t - lo_t, opt)
    assert eng.world == world and eng.fused == (opt == "adabn")      # (use_bn is part of the fused step since round 6; DAN / JAN / MCD are unfused lists)
    shapes = {n: s for n, _, s, _ in eng.plan.params}
    eng.load_state(synth_state(shapes, seed=3))
    res = _run_da(eng, xs[lo:hi], xt[lo_t:hi_t], ys[lo:hi], 2, global_source=c["Bs"], global_target=c["Bt"])
    torch.save(res, f"{out}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("opt", sorted(DA_OPTS))
def test_two_ranks_with_da_options(tmp_path, opt):
    """dis_DA / ens_DA: two ranks end on the parameters of one process that stepped the whole batch (the reference's DataParallel
    takes those losses on the gathered global batch, main.py:446-562).  use_bn: statistics are per replica there, so the two-rank
    result is NOT the one-process result by design - the ranks must agree with each other, move the parameters, and hold replica
    0's running statistics after sync_buffers()."""
    c = CFG
    out = str(tmp_path / "P")
    mp.spawn(_worker_da, args=(2, _free_port(), out, opt), nprocs=2, join=True)
    (p0, b0, d0), (p1, b1, d1) = torch.load(out + ".0"), torch.load(out + ".1")
    assert torch.equal(p0, p1) and torch.equal(b0, b1) and d0 == d1
    assert torch.isfinite(p0).all()
    xs, xt, ys, yt = synth_batch(c["C"], c["T"], c["D"], c["Bs"], c["Bt"], seed=9)
    eng = _engine_da(c["Bs"], c["Bt"], opt)
    shapes = {n: s for n, _, s, _ in eng.plan.params}
    eng.load_state(synth_state(shapes, seed=3))
    start = eng.P.detach().cpu().clone()
    ref, bref, dref = _run_da(eng, xs, xt, ys, 2)
    assert (p0 - start).abs().max() > 1e-4
    if opt == "adabn":
        assert b0.numel() > 0 and not torch.equal(b0, bref)      # half-batch statistics of replica 0, not whole-batch ones
        return
    assert torch.allclose(p0, ref, rtol=2e-4, atol=2e-6), (p0 - ref).abs().max()
    assert abs(d0 - dref) <= 1e-5 * max(abs(dref), 1e-3)
