"""TA3N_FLAG_F32_SPLIT | TA3N_FLAG_BF16_STORE ("pair twins"): the fp32-grade split arithmetic a b ~ a_lo b_hi + a_hi b_lo + a_hi b_hi
with the hi / lo planes STORED by whoever produces the data (GEMM epilogues, heads kernel, optimiser, ta3n_refresh_bf16, the
feature-store gathers) instead of split in every K loop.
  * every plane is bit for bit what splitting the fp32 original gives, after several pipelined updates;
  * the step is the same arithmetic as the in-register split (differences = summation order inside the MFMAs);
  * the fp32 parity bounds of the split arithmetic hold (tests/test_gpu_gradients.py and test_gpu_parity.py run it as "f32x3p")."""
import pytest
import torch

from golden_util import Golden, case_config
from ta3n_amd import feature_store
from ta3n_amd.engine import TrainEngine
from ta3n_amd.synthetic import synth_batch, synth_state

pytestmark = pytest.mark.gpu


def _split(t):
    """(hi bits, lo bits) of x = hi + lo, hi = RNE_bf16(x), lo = RNE_bf16(x - hi), as int16 tensors."""
    t = t.reshape(-1).float()
    hi = t.to(torch.bfloat16)
    lo = (t - hi.float()).to(torch.bfloat16)
    return hi.view(torch.int16), lo.view(torch.int16)


@pytest.mark.parametrize("pipelined", [True, False])
@pytest.mark.parametrize("name", ["headline", "tiny_T9"])
def test_every_plane_is_the_split_of_its_original_after_several_updates(name, pipelined):
    g = Golden(name)
    c = case_config(g)
    T = c["T"]
    res = []
    for pairs in (False, True):
        eng = TrainEngine(c["Bs"], c["Bt"], T, c["D"], c["fc_dim"], c["C"], dropout_i=0.5, dropout_v=0.5, clip=c["clip"],
                          f32_split=True, bf16_store=pairs)
        shapes = {n: s for n, _, s, _ in eng.plan.params}
        eng.load_state(synth_state(shapes, seed=c["wseed"], scale=c["wscale"]))
        for step in range(3):
            xs, xt, ys, yt = synth_batch(c["C"], T, c["D"], c["Bs"], c["Bt"], seed=100 + step)
            eng.set_batch(xs.cuda(), xt.cuda(), ys.cuda())
            if pipelined:      # the update rides in the next step's first launch (EPI_SGD side tasks) behind the shared-FC update launch
                eng.train_step_pipelined([0.75, 0.75, 0.5], 0.003, 0.03)
            else:              # one update launch at the end of the step (sgd_kernel)
                eng.train_step([0.75, 0.75, 0.5], 0.003, 0.03)
        eng.flush()
        torch.cuda.synchronize()
        if pairs:
            def planes(region, n, off=0):
                return (eng.region(region).view(torch.int16)[off:off + n], eng.region(region + "_lo").view(torch.int16)[off:off + n])
            for what, (hi, lo), src in (("parameters (optimiser)", planes("p16", eng.P.numel()), eng.P),
                                        ("input (ta3n_refresh_bf16)", planes("x16", eng.X.numel()), eng.X)):
                want_hi, want_lo = _split(src)
                assert torch.equal(hi, want_hi), f"hi plane: {what}"
                assert torch.equal(lo, want_lo), f"lo plane: {what}"
            for rname in ("F1", "Zr", "gHf"):
                off, size = eng.plan.regions[rname]
                hi, lo = planes("ws16", size, off)
                want_hi, want_lo = _split(eng.region(rname))
                assert torch.equal(hi, want_hi) and torch.equal(lo, want_lo), f"{rname} planes (written by the producing launch)"
            for rname in ("gZ", "gZ1"):      # read by GEMM launches only: no fp32 copy; hi + lo must be a sane split (|lo| <= ulp_bf16(hi) / 2)
                off, size = eng.plan.regions[rname]
                hi, lo = planes("ws16", size, off)
                hi_f, lo_f = hi.view(torch.bfloat16).float(), lo.view(torch.bfloat16).float()
                assert torch.isfinite(hi_f).all() and hi_f.abs().max().item() > 0 and lo_f.abs().max().item() > 0, rname
                assert (lo_f.abs() <= hi_f.abs() * 2.0 ** -8 + 1e-38).all(), rname
                assert eng.region(rname).abs().max().item() == 0, rname
        res.append((eng.P.clone(), eng.region("losses")[:6].clone()))
    (p0, l0), (p1, l1) = res
    scale = p0.abs().max().item()
    d = (p0 - p1).abs()      # same products, other summation order inside the MFMAs (k grouping 4 + 4 vs 8): fp32 round-off only
    assert d.max().item() <= 1e-4 * scale and d.mean().item() <= 2e-6 * scale, (d.max().item(), d.mean().item(), scale)
    assert torch.allclose(l0, l1, rtol=1e-4, atol=1e-5)


def test_bf16_feature_store_feeds_the_hi_plane_and_zeroes_the_lo_plane(tmp_path):
    D, T, Bs, Bt = 512, 5, 6, 4
    gen = torch.Generator().manual_seed(3)
    lines = []
    for v in range(12):
        d = tmp_path / f"v{v}"
        d.mkdir()
        n = 4 + (5 * v) % 11
        for f in range(1, n + 1):
            torch.save(torch.randn(D, generator=gen).abs(), str(d / f"img_{f:05d}.t7"))
        lines.append(f"{d}/ {n} {v % 7}")
    (tmp_path / "l.txt").write_text("\n".join(lines) + "\n")
    for dtype in ("bf16", "f32"):
        feature_store.pack(str(tmp_path / "l.txt"), str(tmp_path / f"store_{dtype}"), dtype=dtype)
    for dtype in ("bf16", "f32"):
        store = feature_store.FeatureStore(str(tmp_path / f"store_{dtype}"), D)
        eng = TrainEngine(Bs, Bt, T, D, 64, 7, f32_split=True, bf16_store=True)
        eng.region("x16_lo").fill_(1.0)      # stale bytes a gather must overwrite
        ids = torch.arange(Bs, dtype=torch.int32).cuda()
        store.gather_into(eng, ids, 0, labels_out=eng._labels[:Bs])
        ref = TrainEngine(Bs, Bt, T, D, 64, 7)      # fp32 engine: the rows themselves
        feature_store.FeatureStore(str(tmp_path / "store_f32"), D).gather_into(ref, ids, 0, labels_out=ref._labels[:Bs])
        torch.cuda.synchronize()
        rows = ref.X.reshape(-1)[:Bs * T * D]
        n = rows.numel()
        hi = eng.region("x16").view(torch.int16)[:n]
        lo = eng.region("x16_lo").view(torch.int16)[:n]
        want_hi, want_lo = _split(rows)
        assert torch.equal(hi, want_hi)
        if dtype == "bf16":
            nz = (lo != 0).nonzero().flatten()      # a bf16 row is its own hi plane
            assert nz.numel() == 0, (store.bf16, nz.numel(), nz[:8].tolist(), nz[-8:].tolist(), lo[nz[:8]].tolist())
        else:
            assert torch.equal(lo, want_lo)
