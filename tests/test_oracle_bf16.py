"""The bf16 arithmetic (BASELINE configs[1]) of the CPU oracle, checked on the CPU:

 * it is the fp32 oracle plus bf16 operand rounding - close to the fp32 goldens (bf16 noise), not equal to them;
 * the numpy execution of the product's launch plan with bf16-rounded contraction operands (tests/plan_interp.py)
   agrees with it tightly.  The oracle's bf16 mode is written against the reference's layer structure
   (oracle/ta3n_oracle.py: BF16_POLICY), the interpreter executes the plan's Seg / Task arrays: two independent
   statements of the same arithmetic.  The -m gpu twin of this test (tests/test_gpu_bf16.py) compares the HIP
   kernels with the same oracle."""
import numpy as np
import pytest
import torch

from golden_util import Golden, case_config, step_schedule
from oracle import ta3n_oracle as orc
from plan_interp import Interp
from ta3n_amd import _lib
from ta3n_amd.synthetic import synth_batch, synth_state
from test_plan_cpu import ALL_FLAGS, make_hyper


def oracle_bf16_step(c, st, params, agg="trn-m", twins=True):
    cfg = orc.Config(num_class=c["C"], num_segments=c["T"], feature_dim=c["D"], fc_dim=c["fc_dim"], dropout_i=0.0,
                     dropout_v=0.0, arithmetic="bf16", frame_aggregation=agg, bf16_twins=twins)
    xs, xt, ys, yt = synth_batch(c["C"], c["T"], c["D"], c["Bs"], c["Bt"], seed=st["xseed"])
    xs[st["n_src"]:] = 0; xt[st["n_tgt"]:] = 0
    state = orc.TrainState(params={k: v.clone() for k, v in params.items()}, lr=st["lr"])
    res = orc.train_step(state, xs, xt, ys, [0.75, 0.75, 0.5], 0.003, cfg, clip=c["clip"], n_src=st["n_src"], n_tgt=st["n_tgt"])
    return res, state, (xs, xt, ys)


def rel_err(got, want):
    got = np.asarray(got, np.float64); want = np.asarray(want, np.float64)
    return np.abs(got - want).max() / (np.sqrt((want * want).mean()) + 1e-30)


def test_rne_bf16_is_round_to_nearest_even():
    x = torch.tensor([1.0, 1.00390625, 1.005859375, 1.01171875, -3.3895313892515355e38, 1e-40, 0.0], dtype=torch.float32)
    r = orc.rne_bf16(x)
    # ties go to the even mantissa: 1 + 2^-8 -> 1.0, 1 + 3 * 2^-8 -> 1 + 2^-6; 1 + 1.5 * 2^-8 rounds up to 1 + 2^-7
    assert r[0] == 1.0 and r[1] == 1.0 and r[2] == 1.0078125 and r[3] == 1.015625
    from plan_interp import round_bf16
    v = torch.randn(4096, dtype=torch.float32) * 37.0
    assert np.array_equal(round_bf16(v.numpy()), orc.rne_bf16(v).numpy())


@pytest.mark.parametrize("name", ["tiny_T5", "tiny_T9", "tiny_T3"])
def test_bf16_oracle_is_near_but_not_equal_to_the_fp32_reference(name):
    g = Golden(name)
    c = case_config(g)
    params = synth_state(orc.param_shapes(orc.Config(num_class=c["C"], num_segments=c["T"], feature_dim=c["D"], fc_dim=c["fc_dim"])),
                         seed=c["wseed"], scale=c["wscale"])
    res, _, _ = oracle_bf16_step(c, step_schedule(c)[0], params)
    for dom, key in (("s", "src"), ("t", "tgt")):
        ref = g.z[f"fwd/out_{dom}#full"].astype(np.float64)
        e = rel_err(res[key]["out"].detach().numpy(), ref)
        assert 1e-5 < e < 5e-2, e          # bf16 operand noise: ~1e-3..1e-2 of rms, never fp32-exact


@pytest.mark.parametrize("tile", [0, 32222, 22222, 12222, 32221])
def test_register_blocked_tiles_only_where_the_launch_reads_twins(tile):
    """Tile codes >= 10000 ask for 2 / 2x2 32x32 blocks per wave (bf16-twin kernel only).  The plan keeps them on launches that
    read twins and silently falls back to one block per wave elsewhere (fp32 arithmetic, launches with odd-shaped operands)."""
    from plan_interp import plan_arrays
    c = case_config(Golden("tiny_T5"))
    for flags in (ALL_FLAGS, ALL_FLAGS | _lib.FLAG_BF16_MFMA, ALL_FLAGS | _lib.FLAG_BF16_MFMA | _lib.FLAG_BF16_STORE):
        plan = _lib.Plan(c["Bs"], c["Bt"], c["T"], c["D"], c["fc_dim"], c["C"], flags, tile_config=tile)
        _, tasks, phases, _, _, _ = plan_arrays(plan)
        blocked = 0
        for ph in phases:
            if ph.kind != 0:
                continue
            rm, rn = max(ph.rm, 1), max(ph.rn, 1)
            if rm * rn > 1:
                assert ph.bf16 >= 16, (tile, flags, ph.bf16)
                blocked += 1
            BM, BN = 32 * ph.wm * rm, 32 * ph.wn * rn
            for t in tasks[ph.task_begin:ph.task_begin + ph.task_count]:
                if t.seg_count > 0:
                    assert t.m0 % BM == 0 and t.n0 % BN == 0
        twins = bool(flags & _lib.FLAG_BF16_STORE)
        assert (blocked > 0) == (twins and tile >= 10000), (tile, flags, blocked)


@pytest.mark.parametrize("store,tile", [(False, 0), (True, 0), (True, 32222), (True, 22222), (True, 12222), (True, 32221)])
@pytest.mark.parametrize("name", ["tiny_T5", "tiny_T9", "tiny_T3", "tiny_T2"])
def test_plan_in_bf16_matches_the_independent_bf16_oracle(name, store, tile):
    g = Golden(name)
    c = case_config(g)
    T = c["T"]
    flags = ALL_FLAGS | _lib.FLAG_BF16_MFMA | (_lib.FLAG_BF16_STORE if store else 0)
    plan = _lib.Plan(c["Bs"], c["Bt"], T, c["D"], c["fc_dim"], c["C"], flags, tile_config=tile)
    it = Interp(plan)
    shapes = {n: s for n, _, s, _ in plan.params}
    params = synth_state(shapes, seed=c["wseed"], scale=c["wscale"])
    it.set_params(params)
    st = step_schedule(c)[0]
    res, state, (xs, xt, ys) = oracle_bf16_step(c, st, params, twins=store)
    it.X = torch.cat((xs, xt), 0).double().numpy().reshape(-1)
    it.labels[:c["Bs"]] = ys.numpy()
    it.hy = make_hyper(c, st, T, st["lr"])
    it.G[:] = 0
    it.run_group(4)
    B, Bs = c["Bs"] + c["Bt"], c["Bs"]
    geo = it.g
    got = dict(out=it.r(geo.o_Y, (B, c["C"])), rel=it.r(geo.o_Pr, (B, T - 1, 2)), vid=it.r(geo.o_Pv, (B, 2)),
               frm=it.r(geo.o_Pf, (B, T, 2)), v=it.r(geo.o_V, (B, 256)), attn=it.r(geo.o_attn, (B, T - 1)))
    for dom, key, sl in (("s", "src", slice(0, Bs)), ("t", "tgt", slice(Bs, B))):
        o = res[key]
        want = dict(out=o["out"], rel=o["pred_domain"][0], vid=o["pred_domain"][1], frm=o["pred_domain"][2], v=o["feat"][1],
                    attn=o["attn"])
        for k, w in want.items():
            e = rel_err(got[k][sl], w.detach().numpy())
            assert e < 2e-4, (name, dom, k, e)     # identical rounding points; the rest is fp32 summation order (interp: fp64)
    want_g = res["grads"]
    got_g = it.get_params(it.G)
    for k, w in want_g.items():
        w = w.numpy().astype(np.float64)
        scale = np.abs(w).max() + 1e-30
        err = np.abs(got_g[k].reshape(w.shape) - w).max()
        # same rounding points; what remains is fp32-vs-fp64 summation plus the rare operand that sits on a bf16 rounding
        # boundary and flips (2^-8 of one term of a 9..10-term sum at these tiny batches)
        assert err <= 4e-3 * scale, (name, k, err, scale)
