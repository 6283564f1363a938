"""CPU validation of the launch plan: the descriptor lists the HIP kernels
consume are executed with numpy (tests/plan_interp.py) and compared with the
golden vectors produced by the reference, for every fixture.  This pins the
wiring (offsets, K-segments, GradReverse scales, epilogues) without a GPU; the
-m gpu tests then only have to pin the kernels themselves."""
import numpy as np
import pytest
import torch

from golden_util import BN_CASES, CASES, Golden, case_config, step_schedule
from plan_interp import Interp
from ta3n_amd import _lib
from ta3n_amd.synthetic import synth_batch, synth_state

ALL_FLAGS = (_lib.FLAG_ADV_RELATION | _lib.FLAG_ADV_VIDEO | _lib.FLAG_ADV_FRAME | _lib.FLAG_ATTN_ENTROPY |
             _lib.FLAG_TRANS_ATTN)
SMALL = [c for c in CASES if c.startswith("tiny") or c == "mid_T12"]


def make_hyper(c, st, T, lr):
    n_s, n_t = st["n_src"], st["n_tgt"]
    return dict(beta=[0.75, 0.75, 0.5], gamma=0.003, lr=lr, momentum=0.9, weight_decay=1e-4, clip=c["clip"],
                p_drop_i=0.0, p_drop_v=0.0, seed_i=1, seed_v=2, inv_n_cls=1.0 / n_s, inv_n_rel=1.0 / ((n_s + n_t) * (T - 1)),
                inv_n_vid=1.0 / (n_s + n_t), inv_n_frm=1.0 / ((n_s + n_t) * T), inv_n_ent=1.0 / (n_s + n_t),
                valid_source=n_s, valid_target=n_t, train=1)


@pytest.mark.parametrize("name,tile,fused", [(n, t, f) for n in SMALL for t in (114, 222, 0) for f in (False, True)
                                             if t == 114 or n in ("tiny_T5", "tiny_T3")])      # (tile variants on two cases)
def test_plan_reproduces_reference(name, tile, fused):
    """fused=False: ta3n_forward / ta3n_loss / ta3n_backward launch lists; fused=True: the ta3n_train_step list."""
    g = Golden(name)
    c = case_config(g)
    T = c["T"]
    plan = _lib.Plan(c["Bs"], c["Bt"], T, c["D"], c["fc_dim"], c["C"], ALL_FLAGS, tile_config=tile)
    it = Interp(plan)
    shapes = {n: s for n, _, s, _ in plan.params}
    it.set_params(synth_state(shapes, seed=c["wseed"], scale=c["wscale"]))
    live = {n for n, _, _, lv in plan.params if lv}
    assert live == set(str(k) for k in g.meta("live"))
    for s, st in enumerate(step_schedule(c)):
        xs, xt, ys, yt = synth_batch(c["C"], T, c["D"], c["Bs"], c["Bt"], seed=st["xseed"])
        xs[st["n_src"]:] = 0; xt[st["n_tgt"]:] = 0
        it.X = torch.cat((xs, xt), 0).double().numpy().reshape(-1)
        it.labels[:c["Bs"]] = ys.numpy()
        it.hy = make_hyper(c, st, T, st["lr"])
        it.G[:] = 0
        if fused:
            assert plan.has_fused_step
            it.run_group(4)
        else:
            it.run_group(0)
        if s == 0:   # forward outputs vs the reference's VideoModel.forward
            B, Bs = c["Bs"] + c["Bt"], c["Bs"]
            geo = it.g
            outs = dict(out=it.r(geo.o_Y, (B, c["C"])), attn=it.r(geo.o_attn, (B, T - 1)),
                        rel=it.r(geo.o_Pr, (B, T - 1, 2)), vid=it.r(geo.o_Pv, (B, 2)), frm=it.r(geo.o_Pf, (B, T, 2)),
                        v=it.r(geo.o_V, (B, 256)), f1=it.r(geo.o_F1, (B, T, geo.F)))
            for dom, sl in (("s", slice(0, Bs)), ("t", slice(Bs, B))):
                g.check(f"fwd/out_{dom}", outs["out"][sl], 5e-5, 2e-5)
                g.check(f"fwd/attn_{dom}", outs["attn"][sl], 5e-5, 2e-5)
                for nm in ("rel", "vid", "frm"):
                    g.check(f"fwd/pd_{dom}_{nm}", outs[nm][sl], 5e-5, 2e-5)
                g.check(f"fwd/feat_{dom}_v", outs["v"][sl], 5e-5, 2e-5)
                g.check(f"fwd/feat_{dom}_f1", outs["f1"][sl], 5e-5, 2e-5)
        if not fused:
            it.run_group(1)
            it.run_group(2)
        raw = it.get_params(it.G)
        it.run_group(3, fused_norm=fused)
        coef = it.ws[it.g.o_grad_norm + 1]
        new = it.get_params()
        for k in shapes:
            if k in live:
                g.check(f"step{s}/clipped_grad/{k}", raw[k] * coef, 1e-4, 2e-5)
            g.check(f"step{s}/param/{k}", new[k], 1e-4, 2e-5)
    if name == "tiny_clip":
        assert it.ws[it.g.o_grad_norm + 1] < 1.0   # the clip branch was exercised


def test_plan_rejects_bad_configs():
    with pytest.raises(ValueError):
        _lib.Plan(4, 4, 1, 512, 64, 12, ALL_FLAGS)                      # trn-m needs >= 2 segments
    with pytest.raises(ValueError):
        _lib.Plan(4, 4, 5, 512, 64, 12, _lib.FLAG_ATTN_ENTROPY | _lib.FLAG_ADV_VIDEO)   # main.py:559-562 quirk
    with pytest.raises(ValueError):
        _lib.Plan(0, 0, 5, 512, 64, 12, ALL_FLAGS)
    with pytest.raises(ValueError):
        _lib.Plan(4, 4, 5, 512, 64, 12, ALL_FLAGS, tile_config=333)
    with pytest.raises(ValueError):
        _lib.Plan(4, 4, 5, 512, 30, 12, ALL_FLAGS | _lib.FLAG_BN_SHARED)      # use_bn: the BatchNorm launches move four columns of a row at a time
    _lib.Plan(4, 4, 5, 512, 30, 12, ALL_FLAGS)                                # (without BatchNorm any fc_dim goes)


def test_param_table_matches_reference_state_dict():
    from oracle import ta3n_oracle as orc
    plan = _lib.Plan(128, 74, 5, 2048, 512, 12, ALL_FLAGS)
    want = orc.param_shapes(orc.Config())
    got = {n: s for n, _, s, _ in plan.params}
    assert got == want
    assert sum(int(np.prod(s)) for n, _, s, lv in plan.params if lv) == 3483416
    assert sum(int(np.prod(s)) for s in got.values()) == 3884836
    offs = [o for _, o, _, _ in plan.params]
    assert all(o % 4 == 0 for o in offs) and offs == sorted(offs)
    assert plan.live_floats % 4 == 0


@pytest.mark.parametrize("store", [False, True])
def test_bf16_plan_on_cpu(store):
    """TA3N_FLAG_BF16_MFMA (+ _STORE): the numpy execution of the plan with bf16-rounded contraction operands stays within
    bf16 rounding of the reference's fp32 goldens, and with twins every re-addressed Seg lands inside a twin region."""
    name = "tiny_T5"
    g = Golden(name)
    c = case_config(g)
    T = c["T"]
    flags = ALL_FLAGS | _lib.FLAG_BF16_MFMA | (_lib.FLAG_BF16_STORE if store else 0)
    plan = _lib.Plan(c["Bs"], c["Bt"], T, c["D"], c["fc_dim"], c["C"], flags)
    assert plan.has_fused_step
    it = Interp(plan)
    twin_phases = [ph for ph in it.phases if ph.group == 4 and ph.kind == 0 and (ph.bf16 & 16)]
    assert len(twin_phases) == (5 if store else 0)
    if store:
        geo = it.g
        lo, hi = geo.o_ws16, geo.o_x16 + (c["Bs"] + c["Bt"]) * T * c["D"] // 2
        assert 0 <= geo.o_ws16 < geo.o_p16 < geo.o_x16 < geo.o_p16b
        for ph in twin_phases:
            for ti in range(ph.task_begin, ph.task_begin + ph.task_count):
                t = it.tasks[ti]
                for si in range(t.seg_begin, t.seg_begin + t.seg_count):
                    s = it.segs[si]
                    for base, off in ((s.a_base, s.a_off), (s.b_base, s.b_off)):
                        if base == 4:      # BASE_P16: parameter twins, relative to the twin region the launch is handed
                            assert 0 <= off < (plan.param_floats + 1) // 2
                        else:
                            assert base == 3 and lo <= off < hi and not (geo.o_p16 <= off < geo.o_x16)
    shapes = {n: s for n, _, s, _ in plan.params}
    it.set_params(synth_state(shapes, seed=c["wseed"], scale=c["wscale"]))
    st = step_schedule(c)[0]
    xs, xt, ys, yt = synth_batch(c["C"], T, c["D"], c["Bs"], c["Bt"], seed=st["xseed"])
    xs[st["n_src"]:] = 0; xt[st["n_tgt"]:] = 0
    it.X = torch.cat((xs, xt), 0).double().numpy().reshape(-1)
    it.labels[:c["Bs"]] = ys.numpy()
    it.hy = make_hyper(c, st, T, st["lr"])
    it.G[:] = 0
    it.run_group(4)
    B, Bs = c["Bs"] + c["Bt"], c["Bs"]
    out = it.r(it.g.o_Y, (B, c["C"]))
    for dom, sl in (("s", slice(0, Bs)), ("t", slice(Bs, B))):
        rms = g.rms(f"fwd/out_{dom}")
        g.check(f"fwd/out_{dom}", out[sl], 0.0, 0.1 * rms, "bf16 operands vs fp32 reference")


@pytest.mark.parametrize("name", ["tiny_T5", "tiny_T3", "tiny_clip"])
def test_pair_twin_plan_on_cpu(name):
    """TA3N_FLAG_F32_SPLIT | _BF16_STORE ("pair twins": every twin has a hi and a lo plane, the split-arithmetic launches read both
    and split nothing in their K loops): the launch lists reproduce the fp32 goldens (the split arithmetic is fp32-grade: modelled
    as exact products), the lo planes mirror the hi regions at ONE displacement, and the launches that qualify read them."""
    g = Golden(name)
    c = case_config(g)
    T = c["T"]
    plan = _lib.Plan(c["Bs"], c["Bt"], T, c["D"], c["fc_dim"], c["C"], ALL_FLAGS | _lib.FLAG_F32_SPLIT | _lib.FLAG_BF16_STORE)
    assert plan.has_fused_step and not plan.has_fused_update
    it = Interp(plan)
    geo = it.g
    regions = plan.regions      # name -> (offset, size) in floats
    assert geo.pair_delta > 0 and geo.pair_delta % 4 == 0
    for hi_name in ("ws16", "p16", "x16", "p16b"):
        assert regions[hi_name + "_lo"][0] - regions[hi_name][0] == geo.pair_delta and regions[hi_name + "_lo"][1] == regions[hi_name][1]
    assert regions["ws16_lo"][0] >= regions["p16b"][0] + regions["p16b"][1]
    gemm = [ph for ph in it.phases if ph.group == 4 and ph.kind == 0]
    assert [ph.bf16 & 48 for ph in gemm] == [48, 48, 48, 32, 48, 48]      # launch 5 (odd-shaped head gradients) splits fp32 operands in registers
    with pytest.raises(ValueError):
        _lib.Plan(c["Bs"], c["Bt"], T, c["D"], c["fc_dim"], c["C"], ALL_FLAGS | _lib.FLAG_F32_SPLIT | _lib.FLAG_BF16_MFMA)
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


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("name", ["tiny_avgpool"])
def test_avgpool_plan_reproduces_reference(name, fused):
    """BASELINE configs[0] (TemPooling / avgpool, source-only): the TA3N_AGG_AVGPOOL launch lists executed with numpy against
    the fixture the reference produced."""
    g = Golden(name)
    c = case_config(g)
    T = c["T"]
    plan = _lib.Plan(c["Bs"], c["Bt"], T, c["D"], c["fc_dim"], c["C"], 0, aggregation=_lib.AGG_AVGPOOL)
    it = Interp(plan)
    shapes = {n: s for n, _, s, _ in plan.params}
    it.set_params(synth_state(shapes, seed=c["wseed"], scale=c["wscale"]))
    live = {n for n, _, _, lv in plan.params if lv}
    assert live == set(str(k) for k in g.meta("live"))
    with pytest.raises(ValueError):      # adversarial / attention options are not built for avgpool
        _lib.Plan(c["Bs"], c["Bt"], T, c["D"], c["fc_dim"], c["C"], ALL_FLAGS, aggregation=_lib.AGG_AVGPOOL)
    for s, st in enumerate(step_schedule(c)):
        xs, xt, ys, yt = synth_batch(c["C"], T, c["D"], c["Bs"], c["Bt"], seed=st["xseed"])
        xs[st["n_src"]:] = 0; xt[st["n_tgt"]:] = 0
        it.X = torch.cat((xs, xt), 0).double().numpy().reshape(-1)
        it.labels[:c["Bs"]] = ys.numpy()
        it.hy = make_hyper(c, st, T, st["lr"])
        it.G[:] = 0
        it.run_group(4 if fused else 0)
        if s == 0:
            B, Bs = c["Bs"] + c["Bt"], c["Bs"]
            geo = it.g
            outs = dict(out=it.r(geo.o_Y, (B, c["C"])), v=it.r(geo.o_V, (B, geo.F)), f1=it.r(geo.o_F1, (B, T, geo.F)))
            for dom, sl in (("s", slice(0, Bs)), ("t", slice(Bs, B))):
                g.check(f"fwd/out_{dom}", outs["out"][sl], 5e-5, 2e-5)
                g.check(f"fwd/attn_{dom}", outs["v"][sl][:, 0], 5e-5, 2e-5)
                g.check(f"fwd/feat_{dom}_v", outs["v"][sl], 5e-5, 2e-5)
                g.check(f"fwd/feat_{dom}_f1", outs["f1"][sl], 5e-5, 2e-5)
        if not fused:
            it.run_group(1)
            it.run_group(2)
        raw = it.get_params(it.G)
        it.run_group(3, fused_norm=fused)
        coef = it.ws[it.g.o_grad_norm + 1]
        new = it.get_params()
        for k in shapes:
            if k in live:
                g.check(f"step{s}/clipped_grad/{k}", raw[k] * coef, 1e-4, 2e-5)
            g.check(f"step{s}/param/{k}", new[k], 1e-4, 2e-5)


@pytest.mark.parametrize("place_adv,use_attn", [(("Y", "Y", "Y"), "TransAttn"), (("Y", "Y", "N"), "TransAttn"), (("N", "Y", "Y"), "TransAttn"),
                                                (("N", "N", "N"), "TransAttn"), (("N", "N", "Y"), "none"), (("Y", "N", "N"), "none"),
                                                (("N", "N", "N"), "none")])
def test_live_parameter_set_follows_the_options_like_autograd_does(place_adv, use_attn):
    """A discriminator whose output feeds no loss keeps grad None in the reference, so SGD skips it (no weight decay either).
    The plan's live prefix (= the all-reduce and optimiser operand) must be exactly the set autograd gives a gradient."""
    from oracle import ta3n_oracle as orc
    from ta3n_amd.engine import flags_from_options
    ent = "attentive_entropy" if (use_attn != "none" and place_adv[0] == "Y" and place_adv[1] == "Y") else "none"
    flags = flags_from_options(place_adv, ent, use_attn, "RevGrad", "uSv")
    plan = _lib.Plan(3, 2, 4, 32, 16, 5, flags)
    live = {n for n, _, _, lv in plan.params if lv}
    offs = [off for _, off, _, lv in plan.params if lv]
    assert max(offs) < plan.live_floats and all(off >= plan.live_floats for _, off, _, lv in plan.params if not lv)
    cfg = orc.Config(num_class=5, num_segments=4, feature_dim=32, fc_dim=16, dropout_i=0.0, dropout_v=0.0, place_adv=place_adv,
                     add_loss_DA=ent, use_attn=use_attn)
    params = synth_state(orc.param_shapes(cfg), seed=3)
    xs, xt, ys, yt = synth_batch(5, 4, 32, 3, 2, seed=4)
    res = orc.train_step(orc.TrainState(params=params), xs, xt, ys, [0.75, 0.75, 0.5], 0.003, cfg)
    assert live == set(res["grads"].keys())


def test_every_task_carries_a_copy_of_its_first_segment():
    """Task.seg0 (what the kernel opens its K loop with) must be the final - twin-re-addressed - segs[seg_begin]."""
    import ctypes as C
    for flags in (ALL_FLAGS, ALL_FLAGS | _lib.FLAG_BF16_MFMA | _lib.FLAG_BF16_STORE):
        it = Interp(_lib.Plan(6, 4, 5, 512, 64, 12, flags))
        n = 0
        for t in it.tasks:
            if t.seg_count > 0:
                assert bytes(t.seg0) == bytes(it.segs[t.seg_begin])
                n += 1
        assert n > 100


@pytest.mark.parametrize("split_k", [2, 4, 6])      # 2: the gradient at the frame features; 4 (round 6): the shared-FC product, its single K segment halved; 6: both
@pytest.mark.parametrize("flags", [ALL_FLAGS, ALL_FLAGS | _lib.FLAG_BF16_MFMA | _lib.FLAG_BF16_STORE])
@pytest.mark.parametrize("name", ["tiny_T5", "tiny_T9"])
def test_split_k_plan_on_cpu(name, flags, split_k):
    """ta3n_config.split_k = 2: every tile of the gradient at the frame features is two tasks, each over part of the K segments, that
    meet through a partial-tile buffer and a ticket.  Structure of the pairs, and - executed in list order AND with every pair's
    halves swapped - the reference's golden vectors."""
    from plan_interp import EPI_SPLITK, PH_GEMM
    g = Golden(name)
    c = case_config(g)
    T = c["T"]
    plan = _lib.Plan(c["Bs"], c["Bt"], T, c["D"], c["fc_dim"], c["C"], flags, split_k=split_k)
    plain = _lib.Plan(c["Bs"], c["Bt"], T, c["D"], c["fc_dim"], c["C"], flags)
    it = Interp(plan)
    part_off, part_n = plan.regions["splitk_part"]
    tick_off, tick_n = plan.regions["splitk_ticket"]
    assert "splitk_part" not in plain.regions
    if flags & _lib.FLAG_BF16_STORE:      # behind the span the twins mirror
        assert part_off >= plan.regions["p16b"][0] + plan.regions["p16b"][1]
    pairs = {}
    for ph in it.phases:
        if ph.kind != PH_GEMM:
            continue
        tile = 32 * ph.wm * max(ph.rm, 1) * 32 * ph.wn * max(ph.rn, 1)
        for i in range(ph.task_begin, ph.task_begin + ph.task_count):
            t = it.tasks[i]
            if t.epi & EPI_SPLITK:
                assert ph.group in (4, 5) and t.c_base == 3 and t.seg_count >= 1 and t.pad[2] in (1, 2)
                assert part_off <= t.pad[0] and t.pad[0] + 2 * tile <= part_off + part_n and tick_off <= t.pad[1] < tick_off + tick_n
                pairs.setdefault((t.pad[0], t.pad[1]), []).append((i, t))
    assert len(pairs) == tick_n > 0
    for (po, to), halves in pairs.items():
        assert len(halves) == 2
        (i0, a), (i1, b) = sorted(halves, key=lambda h: h[1].pad[2])
        assert (a.m0, a.n0, a.c_off, a.m_valid, a.n_valid) == (b.m0, b.n0, b.c_off, b.m_valid, b.n_valid)
        assert a.seg_begin + a.seg_count == b.seg_begin                   # consecutive shares of the tile's K segments
        ka = sum(it.segs[k].klen for k in range(a.seg_begin, a.seg_begin + a.seg_count))
        kb = sum(it.segs[k].klen for k in range(b.seg_begin, b.seg_begin + b.seg_count))
        assert abs(ka - kb) <= max(it.segs[k].klen for k in range(a.seg_begin, b.seg_begin + b.seg_count))
        assert all(it.segs[k].scale_kind == 0 for k in range(b.seg_begin, b.seg_begin + b.seg_count))    # a scaled Seg stays first in the first half
    shapes = {n: s for n, _, s, _ in plan.params}
    live = {n for n, _, _, lv in plan.params if lv}
    bf16 = bool(flags & _lib.FLAG_BF16_MFMA)
    unsplit = None
    for swap in (None, False, True):      # None: the plan without the split (what the split must reproduce up to summation order)
        it = Interp(plain if swap is None else plan)
        if swap:       # the second half of every pair runs first
            idx = {}
            for (po, to), halves in pairs.items():
                (i0, _), (i1, _) = halves
                idx[i0], idx[i1] = i1, i0
            order = list(range(len(it.tasks)))
            tasks = [it.tasks[idx.get(i, i)] for i in order]
            it.tasks = tasks
        it.set_params(synth_state(shapes, seed=c["wseed"], scale=c["wscale"]))
        st = step_schedule(c)[0]
        xs, xt, ys, yt = synth_batch(c["C"], T, c["D"], c["Bs"], c["Bt"], seed=st["xseed"])
        xs[st["n_src"]:] = 0; xt[st["n_tgt"]:] = 0
        it.X = torch.cat((xs, xt), 0).double().numpy().reshape(-1)
        it.labels[:c["Bs"]] = ys.numpy()
        it.hy = make_hyper(c, st, T, st["lr"])
        it.G[:] = 0
        it.run_group(4)
        assert not it.__dict__.get("_split_parts")                        # every partial was picked up
        raw = it.get_params(it.G)
        it.run_group(3, fused_norm=True)
        coef = it.ws[it.g.o_grad_norm + 1]
        if swap is None:
            unsplit = raw
            continue
        for k in shapes:
            if k in live:
                assert np.allclose(raw[k], unsplit[k], rtol=1e-9, atol=1e-12), k      # (the interpreter accumulates in fp64)
                if not bf16:
                    g.check(f"step0/clipped_grad/{k}", raw[k] * coef, 1e-4, 2e-5)


@pytest.mark.parametrize("tile", [6222, 36222, 35221, 7222, 46221, 56221])
def test_half_stage_and_tall_tile_plans_on_cpu(tile):
    """Tile codes of round 4 (half-stage kernels 6xxx / 7xxx; 192x128 / 256x128 tiles 46221 / 56221): the plan tiles the same
    contractions - the numpy execution of its task lists gives the default plan's logits and gradients - marks the twin launches as
    half-stage launches, and puts the tall tiles only on launches whose A operands are K-contiguous (their kernels hold no k-major A
    loop); everywhere else the launch falls back to a tile its kernel exists for."""
    if tile in (7222, 46221, 56221) and not _lib.has_experiments():
        pytest.skip("four half stages and the four-wave tall tiles live in the experiments build (-DTA3N_EXPERIMENTS=1) since round 5")
    g = Golden("tiny_T5")
    c = case_config(g)
    T = c["T"]
    flags = ALL_FLAGS | _lib.FLAG_BF16_MFMA | _lib.FLAG_BF16_STORE
    results = []
    for tc in (0, tile):
        plan = _lib.Plan(c["Bs"], c["Bt"], T, c["D"], c["fc_dim"], c["C"], flags, tile_config=tc)
        it = Interp(plan)
        shapes = {n: s for n, _, s, _ in plan.params}
        it.set_params(synth_state(shapes, seed=c["wseed"], scale=c["wscale"]))
        st = step_schedule(c)[0]
        xs, xt, ys, yt = synth_batch(c["C"], T, c["D"], c["Bs"], c["Bt"], seed=st["xseed"])
        xs[st["n_src"]:] = 0; xt[st["n_tgt"]:] = 0
        it.X = torch.cat((xs, xt), 0).double().numpy().reshape(-1)
        it.labels[:c["Bs"]] = ys.numpy()
        it.hy = make_hyper(c, st, T, st["lr"])
        it.G[:] = 0
        it.run_group(4)
        results.append((plan, it, it.r(it.g.o_Y, (c["Bs"] + c["Bt"], c["C"])).copy(), it.G[: plan.live_floats].copy()))
    (_, _, y0, g0), (plan, it, y1, g1) = results
    assert np.allclose(y0, y1, rtol=1e-9, atol=1e-12) and np.allclose(g0, g1, rtol=1e-9, atol=1e-12)
    assert np.abs(g0).max() > 0
    desc = [ph for ph in plan.description["phases"] if ph["kind"] == 0 and ph["group"] == 4]
    half = [ph for ph in desc if ph["half_stages"]]
    assert half and all(ph["tile"] >= 16000 for ph in half)                   # only launches that read bf16 twins
    want_rm = {46221: 3, 56221: 4}.get(tile)
    for ph, iph in zip(desc, [p_ for p_ in it.phases if p_.kind == 0 and p_.group == 4]):
        a_kmajor = any(it.segs[si].a_kmajor for ti in range(iph.task_begin, iph.task_begin + iph.task_count)
                       for si in range(it.tasks[ti].seg_begin, it.tasks[ti].seg_begin + it.tasks[ti].seg_count))
        if want_rm is None:
            assert ph["rm"] <= 2
        elif ph["rm"] == want_rm:
            assert ph["half_stages"] and not a_kmajor and ph["tile"] % 1000 == 221
        else:
            assert ph["rm"] <= 2 and (a_kmajor or not ph["half_stages"])
    if want_rm is not None:
        assert any(ph["rm"] == want_rm for ph in desc)


def test_third_stage_rule_keeps_the_second_workgroup_on_unfused_launches_only(monkeypatch):
    """bf16-twin kernel, 64x64 tile: a third 128-k stage (96 KB) costs the CU its second resident workgroup.  The unfused launches
    (groups 0-3) take two stages once a launch has more than one tile per CU; the fused step's launches keep the third stage (the
    same rule measured slower under bench.py's protocol: profiles/r04_half_stage_ab.txt)."""
    flags = ALL_FLAGS | _lib.FLAG_BF16_MFMA | _lib.FLAG_BF16_STORE
    plan = _lib.Plan(128, 128, 12, 1024, 512, 12, flags)      # one stream of BASELINE configs[4]: 384 64x64 tiles in the shared-FC product
    ph = [p_ for p_ in plan.description["phases"] if p_["kind"] == 0]
    stages = lambda p_: (p_["tile"] // 1000) & 15
    first_unfused = next(p_ for p_ in ph if p_["group"] == 0)
    first_fused = next(p_ for p_ in ph if p_["group"] == 4)
    assert first_unfused["task_count"] == 384 and first_unfused["tile"] % 1000 == 222 and stages(first_unfused) == 2
    assert first_fused["task_count"] == 384 and first_fused["tile"] % 1000 == 222 and stages(first_fused) == 3


@pytest.mark.parametrize("name", BN_CASES)
def test_fused_plan_with_domain_batchnorm_reproduces_reference(name):
    """use_bn AdaBN / AutoDIAL inside the FUSED step (round 6; models.py:490-543, 569-570): the ta3n_train_step list
    with its two BatchNorm launches - behind the shared-FC product and in front of its weight gradient - and the fused gradient norm
    (the BatchNorm gradients' share comes from the slots the backward launch leaves) against the reference's own trajectory."""
    g = Golden(name)
    c = case_config(g)
    T = c["T"]
    plan = _lib.Plan(c["Bs"], c["Bt"], T, c["D"], c["fc_dim"], c["C"], ALL_FLAGS | _lib.FLAG_BN_SHARED)
    assert plan.has_fused_step
    kinds = [ph["kind"] for ph in plan.description["phases"] if ph["group"] == 4]
    assert kinds == [0, 10, 0, 0, 6, 0, 0, 11, 0], kinds        # GEMM, BN fwd, GEMM, GEMM, heads, GEMM, GEMM, BN bwd, GEMM
    it = Interp(plan)
    shapes = {n: s for n, _, s, _ in plan.params}
    it.set_params(synth_state(shapes, seed=c["wseed"], scale=c["wscale"]))
    live = {n for n, _, _, lv in plan.params if lv}
    assert live == set(str(k) for k in g.meta("live"))
    for s, st in enumerate(step_schedule(c)):
        xs, xt, ys, yt = synth_batch(c["C"], T, c["D"], c["Bs"], c["Bt"], seed=st["xseed"])
        xs[st["n_src"]:] = 0; xt[st["n_tgt"]:] = 0
        it.X = torch.cat((xs, xt), 0).double().numpy().reshape(-1)
        it.labels[:c["Bs"]] = ys.numpy()
        it.hy = make_hyper(c, st, T, st["lr"])
        it.G[:] = 0
        it.run_group(4)
        total = np.sqrt(sum(float((v.astype(np.float64) ** 2).sum()) for k, v in it.get_params(it.G).items() if k in live))
        slots = np.sqrt(it.ws[it.g.o_sumsq:it.g.o_sumsq + it.g.n_sumsq].sum())
        assert abs(slots - total) <= 1e-6 * total, (slots, total)      # the slots hold the WHOLE norm, BatchNorm gradients included
        it.run_group(3, fused_norm=True)
        new = it.get_params()
        for k in shapes:
            g.check(f"step{s}/param/{k}", new[k], 1e-4, 2e-5)
    # with bf16 twins the launch behind the BatchNorm reads F1's twin (the BatchNorm launch keeps it), the last one gZ0's
    p16 = _lib.Plan(c["Bs"], c["Bt"], T, c["D"], c["fc_dim"], c["C"], ALL_FLAGS | _lib.FLAG_BN_SHARED | _lib.FLAG_BF16_MFMA | _lib.FLAG_BF16_STORE)
    ph4 = [ph for ph in p16.description["phases"] if ph["group"] == 4 and ph["kind"] == 0]
    assert p16.has_fused_step and len(ph4) == 6


def _tile_keys(plan):
    it = Interp(plan)
    out = []
    for ph in it.phases:
        if ph.kind != 0:
            continue
        ts = it.tasks[ph.task_begin:ph.task_begin + ph.task_count]
        # (EPI_SUMROWS8 = 64: the loss-scalar side job rides on whichever tile comes last - order-dependent by design)
        out.append(sorted((t.c_base, t.c_off, t.m0, t.n0, t.seg_begin, t.seg_count, t.epi & ~64) for t in ts if t.seg_count > 0 or t.epi))
    return out


@pytest.mark.parametrize("shape", [(128, 74, 5, 12), (48, 40, 9, 30)])
def test_affinity_group_tile_order_is_a_permutation_of_the_default_order(shape):
    """xcd_aware 3 (round 6): specs that share operand slabs keep their tiles on one XCD, back to back.  Only the
    ORDER of a launch's tiles changes: the same tiles, each exactly once, in every launch - and the numbers the launch lists compute
    are the reference's (tiny_T5 through the interpreter)."""
    Bs, Bt, T, Cn = shape
    a = _lib.Plan(Bs, Bt, T, 2048, 512, Cn, ALL_FLAGS | _lib.FLAG_BF16_MFMA | _lib.FLAG_BF16_STORE, xcd_aware=0)
    b = _lib.Plan(Bs, Bt, T, 2048, 512, Cn, ALL_FLAGS | _lib.FLAG_BF16_MFMA | _lib.FLAG_BF16_STORE, xcd_aware=3)
    ka, kb = _tile_keys(a), _tile_keys(b)
    assert len(ka) == len(kb)
    for x, y in zip(ka, kb):
        assert x == y
    g = Golden("tiny_T5")
    c = case_config(g)
    plan = _lib.Plan(c["Bs"], c["Bt"], c["T"], c["D"], c["fc_dim"], c["C"], ALL_FLAGS, xcd_aware=3)
    it = Interp(plan)
    shapes = {n: s for n, _, s, _ in plan.params}
    it.set_params(synth_state(shapes, seed=c["wseed"], scale=c["wscale"]))
    st = step_schedule(c)[0]
    xs, xt, ys, yt = synth_batch(c["C"], c["T"], c["D"], c["Bs"], c["Bt"], seed=st["xseed"])
    xs[st["n_src"]:] = 0; xt[st["n_tgt"]:] = 0
    it.X = torch.cat((xs, xt), 0).double().numpy().reshape(-1)
    it.labels[:c["Bs"]] = ys.numpy()
    it.hy = make_hyper(c, st, c["T"], st["lr"])
    it.G[:] = 0
    it.run_group(4)
    it.run_group(3, fused_norm=True)
    new = it.get_params()
    for k in shapes:
        g.check(f"step0/param/{k}", new[k], 1e-4, 2e-5)


def test_unfused_lists_read_twins_on_cpu(monkeypatch):
    """Round 6: with TA3N_FLAG_BF16_STORE the UNFUSED lists (ta3n_forward / ta3n_backward - what the DA options with a loss term between
    forward and backward run) read bf16 twins too, analysed as a launch family of their own.  The numpy execution of groups 0, 1, 2 with and
    without them (TA3N_UNFUSED_TWINS=0: fp32 stages rounded in registers, the lists of the rounds before): every weight gradient is the SAME
    number (same operands: RNE of the same fp32 values), the bias gradients that ride on a tile's own A operand (EPI_ROWSUM_A) are sums of the
    rounded values instead of the fp32 ones - as in the fused step - and stay within bf16 rounding.  TA3N_FLAG_MCD plans (no fused step) get
    the twins as well (the engine copies the parameter / input twins into the second pass's workspace)."""
    g = Golden("tiny_T5")
    c = case_config(g)
    T = c["T"]
    flags = ALL_FLAGS | _lib.FLAG_BF16_MFMA | _lib.FLAG_BF16_STORE
    st = step_schedule(c)[0]
    xs, xt, ys, yt = synth_batch(c["C"], T, c["D"], c["Bs"], c["Bt"], seed=st["xseed"])
    grads, plans = {}, {}
    for env in ("1", "0"):
        monkeypatch.setenv("TA3N_UNFUSED_TWINS", env)
        plan = plans[env] = _lib.Plan(c["Bs"], c["Bt"], T, c["D"], c["fc_dim"], c["C"], flags)
        it = Interp(plan)
        unfused_twin = [ph for ph in it.phases if ph.group in (0, 2) and ph.kind == 0 and (ph.bf16 & 16)]
        assert (len(unfused_twin) >= 5) if env == "1" else not unfused_twin, len(unfused_twin)      # shared FC, TRN tuples, relation hidden, gradient at F1 + TRN weight gradients, dWsh
        assert len([ph for ph in it.phases if ph.group == 4 and ph.kind == 0 and (ph.bf16 & 16)]) == 5      # (the fused step's family is analysed on its own)
        it.set_params(synth_state({n: s for n, _, s, _ in plan.params}, seed=c["wseed"], scale=c["wscale"]))
        it.X = torch.cat((xs, xt), 0).double().numpy().reshape(-1)
        it.labels[:c["Bs"]] = ys.numpy()
        it.hy = make_hyper(c, st, T, st["lr"])
        it.G[:] = 0
        for gr in (0, 1, 2):
            it.run_group(gr)
        grads[env] = np.array(it.G[:plan.live_floats], dtype=np.float64)
        if env == "1":
            B, Bs = c["Bs"] + c["Bt"], c["Bs"]
            out = it.r(it.g.o_Y, (B, c["C"]))
            for dom, sl in (("s", slice(0, Bs)), ("t", slice(Bs, B))):
                g.check(f"fwd/out_{dom}", out[sl], 0.0, 0.1 * g.rms(f"fwd/out_{dom}"), "bf16 operands vs fp32 reference")
    for name, off, shape, live in plans["1"].params:
        k = int(np.prod(shape))
        if off + k > len(grads["1"]):
            continue
        a_, b_ = grads["1"][off:off + k], grads["0"][off:off + k]
        if name.endswith(".weight"):
            assert np.array_equal(a_, b_), name
        else:
            assert np.linalg.norm(a_ - b_) <= 5e-3 * np.linalg.norm(b_) + 1e-12, (name, np.linalg.norm(a_ - b_) / np.linalg.norm(b_))
    monkeypatch.setenv("TA3N_UNFUSED_TWINS", "1")
    mcd = Interp(_lib.Plan(c["Bs"], c["Bt"], T, c["D"], c["fc_dim"], c["C"], flags | _lib.FLAG_MCD))      # (no fused step: the unfused lists are all there is)
    assert len([ph for ph in mcd.phases if ph.group in (0, 2) and ph.kind == 0 and (ph.bf16 & 16)]) >= 5
