"""ta3n_train_steps (several pipelined steps enqueued by ONE C call, include/ta3n_hip.h): bit-identical to the same steps
enqueued one call at a time, with the input static in HBM and with the batches assembled on the device from packed stores."""
import pytest
import torch

from golden_util import Golden, case_config
from ta3n_amd import feature_store
from ta3n_amd.engine import TrainEngine
from ta3n_amd.synthetic import synth_batch, synth_state

pytestmark = pytest.mark.gpu


def _engine(c, **kw):
    return TrainEngine(c["Bs"], c["Bt"], c["T"], c["D"], c["fc_dim"], c["C"], dropout_i=0.5, dropout_v=0.5, **kw)


def _load(eng, seed=7):
    shapes = {n: s for n, _, s, _ in eng.plan.params}
    eng.load_state(synth_state(shapes, seed=seed))


@pytest.mark.parametrize("arith", ["f32", "bf16"])
def test_one_call_for_many_steps_is_bit_identical_to_one_call_per_step(arith):
    c = case_config(Golden("tiny_T5"))
    kw = dict(bf16=True, bf16_store=True) if arith == "bf16" else {}
    sched = [([0.1 * (i + 1), 0.75, 0.5], 0.003, 1e-3 * (i + 1)) for i in range(7)]
    results = []
    for mode in ("per_step", "one_call", "two_calls"):
        eng = _engine(c, **kw)
        _load(eng)
        xs, xt, ys, yt = synth_batch(c["C"], c["T"], c["D"], c["Bs"], c["Bt"], seed=11)
        eng.set_batch(xs.cuda(), xt.cuda(), ys.cuda())
        if mode == "per_step":
            for b, g, lr in sched:
                eng.train_step_pipelined(b, g, lr)
        elif mode == "one_call":
            eng.train_steps(sched)
        else:
            eng.train_steps(sched[:3])
            eng.train_steps(sched[3:])
        eng.flush()
        torch.cuda.synchronize()
        assert eng.step_count == len(sched)
        results.append((eng.P.clone(), eng.M.clone(), eng.region("losses")[:6].clone()))
    for other in results[1:]:
        assert torch.equal(results[0][0], other[0]) and torch.equal(results[0][1], other[1]) and torch.equal(results[0][2], other[2])
    assert torch.isfinite(results[0][0]).all()


def _make_store(tmp_path, name, n_videos, D, dtype):
    g = torch.Generator().manual_seed(5 + n_videos)
    lines = []
    for v in range(n_videos):
        d = tmp_path / f"{name}{v}"
        d.mkdir()
        n = 3 + (7 * v) % 23
        for f in range(1, n + 1):
            torch.save(torch.randn(D, generator=g).abs(), str(d / f"img_{f:05d}.t7"))
        lines.append(f"{d}/ {n} {v % 7}")
    lst = tmp_path / f"{name}.txt"
    lst.write_text("\n".join(lines) + "\n")
    prefix = str(tmp_path / name)
    feature_store.pack(str(lst), prefix, dtype=dtype)
    return feature_store.FeatureStore(prefix, D)


@pytest.mark.parametrize("arith,store_dtype", [("f32", "f32"), ("bf16", "f32"), ("bf16", "bf16")])
def test_device_side_batch_feeds_match_per_step_gathers(tmp_path, arith, store_dtype):
    Bs, Bt, T, D, F, C = 6, 4, 5, 512, 64, 7
    src, tgt = _make_store(tmp_path, "s", 14, D, store_dtype), _make_store(tmp_path, "t", 9, D, store_dtype)
    kw = dict(bf16=True, bf16_store=True) if arith == "bf16" else {}
    n = 5
    gen = torch.Generator().manual_seed(3)
    ids_s = torch.stack([torch.randperm(14, generator=gen)[:Bs] for _ in range(n)]).to(torch.int32).cuda()
    ids_t = torch.stack([torch.randperm(9, generator=gen)[:Bt] for _ in range(n)]).to(torch.int32).cuda()
    sched = [([0.75, 0.75, 0.5], 0.003, 2e-3) for _ in range(n)]
    results = []
    for mode in ("per_step", "one_call"):
        eng = TrainEngine(Bs, Bt, T, D, F, C, dropout_i=0.5, dropout_v=0.5, **kw)
        _load(eng)
        if mode == "per_step":
            for k, (b, g, lr) in enumerate(sched):
                src.gather_into(eng, ids_s[k], 0, labels_out=eng._labels[:Bs])
                tgt.gather_into(eng, ids_t[k], Bs)
                eng.train_step_pipelined(b, g, lr)
        else:
            eng.train_steps(sched, feeds=((src, ids_s), (tgt, ids_t)))
        eng.flush()
        torch.cuda.synchronize()
        results.append((eng.P.clone(), eng.M.clone(), eng.region("losses")[:6].clone(), eng._labels[:Bs].clone()))
    a, b = results
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    assert torch.isfinite(a[0]).all() and a[2][0].item() > 0
