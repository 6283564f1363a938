"""Chained launches (ta3n_config.chain: several GEMM dependency levels in ONE launch with tile-level hand-offs) on the CPU:
the launch lists are executed by the numpy interpreter
  * in list order - must reproduce the reference's golden vectors like the unchained plan does, and
  * in ADVERSARIAL orders that respect only the declared hand-offs (every task as early as its wait list allows, highest index
    first; and random orders) - must give the same numbers: a wait list that misses a producer makes its consumer read stale
    data in such an order.
Structural checks: a task only waits on lower-indexed tasks' counters, counters reach their targets exactly, the launch count."""
import numpy as np
import pytest
import torch

from golden_util import Golden, case_config, step_schedule
from plan_interp import PH_GEMM, Interp, plan_waits
from test_plan_cpu import ALL_FLAGS, make_hyper
from ta3n_amd import _lib
from ta3n_amd.synthetic import synth_batch, synth_state


def _chained(plan):
    it = Interp(plan)
    return [ph for ph in it.phases if ph.kind == PH_GEMM and ph.chain_off >= 0]


@pytest.mark.parametrize("name", ["tiny_T5", "tiny_T3", "tiny_T9", "mid_T12"])
def test_chained_plan_reproduces_reference(name):
    g = Golden(name)
    c = case_config(g)
    T = c["T"]
    plan = _lib.Plan(c["Bs"], c["Bt"], T, c["D"], c["fc_dim"], c["C"], ALL_FLAGS, chain=1)
    assert [ph["kind"] for ph in plan.description["phases"] if ph["group"] == 4] == [0, 6, 0, 0]      # 4 launches + the update = 5 per step
    it = Interp(plan)
    shapes = {n: s for n, _, s, _ in plan.params}
    it.set_params(synth_state(shapes, seed=c["wseed"], scale=c["wscale"]))
    live = {n for n, _, _, lv in plan.params if lv}
    for s, st in enumerate(step_schedule(c)):
        xs, xt, ys, yt = synth_batch(c["C"], T, c["D"], c["Bs"], c["Bt"], seed=st["xseed"])
        xs[st["n_src"]:] = 0; xt[st["n_tgt"]:] = 0
        it.X = torch.cat((xs, xt), 0).double().numpy().reshape(-1)
        it.labels[:c["Bs"]] = ys.numpy()
        it.hy = make_hyper(c, st, T, st["lr"])
        it.G[:] = 0
        it.run_group(4)
        raw = it.get_params(it.G)
        it.run_group(3, fused_norm=True)
        coef = it.ws[it.g.o_grad_norm + 1]
        new = it.get_params()
        for k in shapes:
            if k in live:
                g.check(f"step{s}/clipped_grad/{k}", raw[k] * coef, 1e-4, 2e-5)
            g.check(f"step{s}/param/{k}", new[k], 1e-4, 2e-5)


def _run_step(plan, c, order_mode, seed=0, pipelined=False):
    """One fused step; chained launches run in list order (order_mode None) or in a hand-off-respecting adversarial / random order."""
    it = Interp(plan)
    shapes = {n: s for n, _, s, _ in plan.params}
    it.set_params(synth_state(shapes, seed=7))
    xs, xt, ys, yt = synth_batch(c["C"], c["T"], c["D"], c["Bs"], c["Bt"], seed=3)
    it.X = torch.cat((xs, xt), 0).double().numpy().reshape(-1)
    it.labels[:c["Bs"]] = ys.numpy()
    st = dict(n_src=c["Bs"], n_tgt=c["Bt"])
    it.hy = make_hyper(dict(clip=20.0), st, c["T"], 1e-2)
    rng = np.random.default_rng(5)
    if pipelined:      # a previous step's gradients / norm partials, applied by the side tasks of the first launch
        it.G[:it.g.live_floats] = rng.standard_normal(it.g.live_floats) * 1e-2
        it.M[:it.g.live_floats] = rng.standard_normal(it.g.live_floats) * 1e-2
        it.ws[it.g.o_sumsq:it.g.o_sumsq + it.g.n_sumsq] = rng.random(it.g.n_sumsq)
        it.side = dict(lr=0.05, momentum=0.9, weight_decay=1e-4, clip=20.0)
    groups = (5, 4) if pipelined else (4,)
    first = True
    for grp in groups:
        phs = [ph for ph in it.phases if ph.group == grp]
        if grp == 4 and pipelined:
            phs = phs[1:]                                     # ta3n_train_step_after_update: group 5, then group 4 without its first launch
        for ph in phs:
            if ph.kind == PH_GEMM and ph.chain_off >= 0 and order_mode is not None:
                it.run_gemm(ph, it.chain_order(ph, adversarial=(order_mode == "adversarial"), seed=seed))
            elif ph.kind == PH_GEMM:
                it.run_gemm(ph)
            else:
                it.run_heads()
    return it


@pytest.mark.parametrize("pipelined", [False, True])
@pytest.mark.parametrize("shape", [dict(Bs=6, Bt=4, T=5, D=512, F=64, C=12), dict(Bs=40, Bt=30, T=3, D=256, F=128, C=7),
                                   dict(Bs=33, Bt=37, T=9, D=192, F=64, C=30)])
def test_declared_handoffs_cover_every_dependency(shape, pipelined):
    c = shape
    plan = _lib.Plan(c["Bs"], c["Bt"], c["T"], c["D"], c["F"], c["C"], ALL_FLAGS, chain=1, tile_config=124)
    ref = _run_step(plan, c, None, pipelined=pipelined)
    for mode, seed in (("adversarial", 0), ("random", 1), ("random", 2)):
        it = _run_step(plan, c, mode, seed, pipelined=pipelined)
        assert np.array_equal(it.ws, ref.ws), mode
        assert np.array_equal(it.G, ref.G) and np.array_equal(it.P, ref.P) and np.array_equal(it.M, ref.M), mode


@pytest.mark.parametrize("flags", [ALL_FLAGS, ALL_FLAGS | _lib.FLAG_BF16_MFMA | _lib.FLAG_BF16_STORE])
def test_chain_structure_at_the_headline_shape(flags):
    from ta3n_amd.tuning import tuned_phase_tiles
    bf16 = bool(flags & _lib.FLAG_BF16_MFMA)
    plan = _lib.Plan(128, 74, 5, 2048, 512, 12, flags, chain=1, phase_tiles=tuned_phase_tiles(202, 5, 2048, 512, bf16, bf16))
    it = Interp(plan)
    waits = plan_waits(plan)
    chained = [ph for ph in it.phases if ph.kind == PH_GEMM and ph.chain_off >= 0]
    assert len(chained) == 3                              # forward, backward tail, forward with the update riding in it
    for ph in chained:
        ids = range(ph.task_begin, ph.task_begin + ph.task_count)
        signalled = np.zeros(ph.chain_n, np.int64)
        first_sig = {}
        for i in ids:
            t = it.tasks[i]
            for w in range(t.wait_begin, t.wait_begin + t.wait_count):
                cnt, target = waits[w]
                assert 0 <= cnt < ph.chain_n
                # every producer of that counter precedes the waiting task in the launch (workgroups are dispatched in index order)
                assert signalled[cnt] == target, (i, cnt, signalled[cnt], target)
            if t.sig >= 0:
                signalled[t.sig] += 1
                first_sig.setdefault(t.sig, i)
        assert (signalled > 0).all()
        order = it.chain_order(ph)                        # the adversarial order exists: no deadlock
        assert sorted(order) == list(ids)
    # the same tile arithmetic as the unchained plan where the tile shape is the same: identical task lists up to order
    plain = _lib.Plan(128, 74, 5, 2048, 512, 12, flags, chain=0, phase_tiles=tuned_phase_tiles(202, 5, 2048, 512, bf16, bf16))
    assert plan.param_floats == plain.param_floats and plan.live_floats == plain.live_floats
