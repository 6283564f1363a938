"""RCCL from the C ABI (include/ta3n_hip.h: ta3n_comm_*, ta3n_all_reduce_sum, ta3n_train_step_ddp).

On a 1-GPU box the data-parallel code path runs in a 1-rank communicator (the transfer is the identity, everything else -
dlopen of RCCL, ncclCommInitRank, ncclAllReduce on the step's stream, the two-stream schedule with its event edges, the bf16
transport kernels - is the real thing).  With two or more GPUs visible the 2-rank test runs real ranks over xGMI."""
import ctypes as C
import os
import socket

import pytest
import torch

from ta3n_amd.synthetic import synth_batch, synth_state

pytestmark = pytest.mark.gpu
CFG = dict(C=12, T=5, D=512, fc=128, Bs=12, Bt=8)


def _engine(monkeypatch, bf16=False, buckets=1, selftest=True, transport=None):
    from ta3n_amd.engine import TrainEngine
    if selftest:
        monkeypatch.setenv("TA3N_DDP_SELFTEST", "1")
    else:
        monkeypatch.delenv("TA3N_DDP_SELFTEST", raising=False)
    monkeypatch.setenv("TA3N_DDP_BUCKETS", str(buckets))
    if transport is not None:
        monkeypatch.setenv("TA3N_DDP_BF16", transport)
    c = CFG
    eng = TrainEngine(c["Bs"], c["Bt"], c["T"], c["D"], c["fc"], c["C"], dropout_i=0.0, dropout_v=0.0, bf16=bf16, bf16_store=bf16)
    eng.load_state(synth_state({n: s for n, _, s, _ in eng.plan.params}, seed=3))
    return eng


def _steps(eng, n=3):
    c = CFG
    xs, xt, ys, yt = synth_batch(c["C"], c["T"], c["D"], c["Bs"], c["Bt"], seed=9)
    for i in range(n):
        eng.set_batch(xs.cuda(), xt.cuda(), ys.cuda())
        eng.train_step([0.75, 0.75, 0.5], 0.003, 1e-2, seed=i)
    eng.flush()
    torch.cuda.synchronize()
    return eng.P.detach().clone()


@pytest.mark.parametrize("buckets", [1, 2])
def test_one_rank_rccl_step_equals_the_plain_step(monkeypatch, buckets):
    """fp32 transport in a 1-rank communicator is the identity: the data-parallel sequence (step, ncclAllReduce on the step's
    stream or on the second stream beside the last launch, gradient-norm pass, update) must give the plain step's parameters."""
    ref = _steps(_engine(monkeypatch, selftest=False))
    eng = _engine(monkeypatch, buckets=buckets, transport="0")
    assert eng.comm is not None and eng.comm.world == 1 and eng._g16 is None
    got = _steps(eng)
    # same gradients; the norm comes from a pass over the buffer instead of the per-tile partials (another summation order)
    assert torch.allclose(got, ref, rtol=1e-5, atol=1e-7), (got - ref).abs().max()
    assert not torch.equal(got, synth_flat(eng))


# (opt-in exchange TA3N_DDP_SHARDED=1: shipped in the default library and probed by bench.py at N > 1, so part of `-m gpu`)
@pytest.mark.parametrize("how", ["per_step", "pipelined", "one_call_two_streams", "one_call_one_stream", "bf16_arithmetic"])
def test_one_rank_sharded_update_equals_the_plain_step(monkeypatch, how):
    """TA3N_DDP_SHARDED=1: reduce-scatter -> per-shard sum of squares -> all-gather of one float per rank -> clip + SGD on the own
    shards -> all-gather of the parameters (ta3n_sharded_update / ta3n_train_steps_sharded), in a 1-rank communicator: the
    collectives are identities, the shard bookkeeping, the event edges between the two streams and the twin refresh are real.  Same
    gradients as the plain step; the norm is summed in another order (allclose, as for the all-reduce path)."""
    bf16 = how == "bf16_arithmetic"
    ref = _steps(_engine(monkeypatch, selftest=False, bf16=bf16), n=5)
    monkeypatch.setenv("TA3N_DDP_SHARDED", "1")
    monkeypatch.setenv("TA3N_DDP_SHARDED_STREAMS", "1" if how == "one_call_one_stream" else "2")
    eng = _engine(monkeypatch, transport="0", bf16=bf16)
    assert eng._sharded and eng.comm is not None and eng._shard_own == [0, eng._n_first, eng._shard_layout[1], eng.plan.live_floats]
    c = CFG
    xs, xt, ys, yt = synth_batch(c["C"], c["T"], c["D"], c["Bs"], c["Bt"], seed=9)
    eng.set_batch(xs.cuda(), xt.cuda(), ys.cuda())
    sched = [([0.75, 0.75, 0.5], 0.003, 1e-2)] * 5
    if how == "per_step":
        for i, (b, g, lr) in enumerate(sched):
            eng.train_step(b, g, lr, seed=i)
    elif how == "pipelined":
        for i, (b, g, lr) in enumerate(sched):
            eng.train_step_pipelined(b, g, lr, seed=i)
    else:
        eng.train_steps(sched[:2])
        eng.train_steps(sched[2:])
    eng.flush()
    torch.cuda.synchronize()
    assert torch.allclose(eng.P, ref, rtol=1e-5, atol=1e-7), (eng.P - ref).abs().max()
    if bf16:      # the twins of the gathered parameters were rebuilt locally: bit for bit the rounding of the fp32 parameters
        off, n = eng.plan.region("p16")
        tw = eng.ws[off:off + n].view(torch.bfloat16)[: eng.plan.param_floats]
        assert torch.equal(tw, eng.P.to(torch.bfloat16))


def test_bf16_transport_rounds_each_ranks_gradient_to_bf16(monkeypatch):
    assert _engine(monkeypatch, bf16=True, transport=None)._g16 is None      # default: fp32 transport in every arithmetic
    eng = _engine(monkeypatch, bf16=True, transport="1")           # opt-in (TA3N_DDP_BF16=1 / TrainEngine(grad_transport="bf16"))
    assert eng._g16 is not None
    c = CFG
    xs, xt, ys, yt = synth_batch(c["C"], c["T"], c["D"], c["Bs"], c["Bt"], seed=9)
    eng.set_batch(xs.cuda(), xt.cuda(), ys.cuda())
    eng.set_hyper([0.75, 0.75, 0.5], 0.003, 1e-2)
    eng.fused_step()
    g = eng.G[: eng.plan.live_floats].clone()
    eng.all_reduce_grads()
    torch.cuda.synchronize()
    want = g.to(torch.bfloat16).to(torch.float32)                    # round to nearest even, one rank: the sum is the value itself
    assert torch.equal(eng.G[: eng.plan.live_floats], want) and g.abs().max().item() > 0


def test_raw_all_reduce_entry_point(monkeypatch):
    from ta3n_amd import _lib, parallel
    L = _lib.lib()
    comm = parallel.NativeComm(None, torch.device("cuda", 0))
    x = torch.randn(4096, device="cuda")
    y = x.clone()
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(L.ta3n_all_reduce_sum(comm.handle, y.data_ptr(), y.numel(), None, s), "ta3n_all_reduce_sum")
    torch.cuda.synchronize()
    assert torch.equal(x, y) and L.ta3n_comm_world(comm.handle) == 1
    assert L.ta3n_all_reduce_sum(comm.handle, y.data_ptr(), 6, y.data_ptr(), s) < 0      # bf16 transport needs count % 4 == 0
    comm.close()


def synth_flat(eng):
    from ta3n_amd.engine import TrainEngine
    e2 = TrainEngine(eng.Bs, eng.Bt, eng.T, eng.D, CFG["fc"], eng.C, dropout_i=0.0, dropout_v=0.0)
    e2.load_state(synth_state({n: s for n, _, s, _ in e2.plan.params}, seed=3))
    return e2.P.detach().clone()


# ---- real ranks (needs >= 2 GPUs) ----
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out, buckets):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      TA3N_DDP_BUCKETS=str(buckets), TA3N_DDP_BF16="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from ta3n_amd import parallel
    from ta3n_amd.engine import TrainEngine
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    c = CFG
    xs, xt, ys, yt = synth_batch(c["C"], c["T"], c["D"], c["Bs"], c["Bt"], seed=9)
    lo, hi = parallel.shard_range(c["Bs"], world, rank)
    lo_t, hi_t = parallel.shard_range(c["Bt"], world, rank)
    eng = TrainEngine(hi - lo, hi_t - lo_t, c["T"], c["D"], c["fc"], c["C"], dropout_i=0.0, dropout_v=0.0)
    assert eng.comm is not None and eng.comm.world == world
    eng.load_state(synth_state({n: s for n, _, s, _ in eng.plan.params}, seed=3))
    for i in range(3):
        eng.set_batch(xs[lo:hi].cuda(), xt[lo_t:hi_t].cuda(), ys[lo:hi].cuda())
        eng.train_step([0.75, 0.75, 0.5], 0.003, 1e-2, seed=i, global_source=c["Bs"], global_target=c["Bt"])
    torch.cuda.synchronize()
    torch.save(eng.P.detach().cpu(), f"{out}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)        # (pytest-timeout: a rendezvous that never completes must not hold the whole GPU tier)
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the round-end scaling run exercises this path on 8)")
@pytest.mark.parametrize("buckets", [1, 2])
def test_two_rccl_ranks_equal_single_process_global_batch(tmp_path, monkeypatch, buckets):
    import torch.multiprocessing as mp
    out = str(tmp_path / "P")
    mp.spawn(_worker, args=(2, _free_port(), out, buckets), nprocs=2, join=True)
    p0, p1 = torch.load(out + ".0"), torch.load(out + ".1")
    ref = _steps(_engine(monkeypatch, selftest=False)).cpu()
    assert torch.equal(p0, p1)
    assert torch.allclose(p0, ref, rtol=2e-4, atol=2e-6), (p0 - ref).abs().max()
