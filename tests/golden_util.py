"""Helpers shared by the parity tests: golden-fixture access and comparison.
Fixture format is defined by tests/golden/make_golden.py."""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
N_SAMP = 64
CASES = ["tiny_T5", "tiny_T3", "tiny_T9", "tiny_T2", "tiny_clip", "headline", "headline_init", "mid_T12"]
AVG_CASES = ["tiny_avgpool", "config1_avgpool"]      # BASELINE configs[0]: TemPooling (avgpool), source-only, every DA option off
AVG_DA_CASES = ["tiny_avgpool_da", "tiny_avgpool_da3", "tiny_avgpool_dav", "tempooling_da"]   # TemPooling + RevGrad (place_adv in the fixture)
DA_EXTRA_CASES = ["tiny_dan", "tiny_dan_all", "tiny_jan", "tiny_mcd", "tiny_mcd_noent", "mid_dan_mcd"]   # dis_DA DAN / JAN, ens_DA MCD on top of TA3N
AVG_DA_EXTRA_CASES = ["tiny_avgpool_dan_mcd", "tiny_avgpool_jan", "tiny_avgpool_adabn", "tiny_avgpool_mcd_noent"]     # the same options on TemPooling
BN_CASES = ["tiny_adabn", "tiny_autodial", "mid_adabn"]      # use_bn AdaBN / AutoDIAL: domain-specific BatchNorm after the shared FC
ARCH_DIM = dict(resnet18=512, resnet34=512, resnet50=2048, resnet101=2048, resnet152=2048)


def sample_index(numel):
    return (np.arange(N_SAMP, dtype=np.int64) * 7919 + 13) % numel


class Golden:
    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)

    def meta(self, k):
        v = self.z["meta/" + k]
        return v.item() if v.shape == () else (v[0].item() if v.shape == (1,) else v)

    def has(self, k):
        return (k + "#full") in self.z or (k + "#stats") in self.z

    def rms(self, key):
        if key + "#full" in self.z:
            r = self.z[key + "#full"].astype(np.float64)
            return float(np.sqrt((r * r).mean())) if r.size else 0.0
        n = int(np.prod(self.z[key + "#shape"]))
        return float(np.sqrt(self.z[key + "#stats"][2] / max(n, 1)))

    def check(self, key, t, rtol, atol, what="", rms_atol=0.0):
        """Compare tensor/array `t` with the stored record `key`.  rms_atol adds
        rms_atol * rms(reference tensor) to the absolute tolerance (for gradients
        after a parameter update, where a ReLU unit flipping on a 1e-7 difference
        moves single entries by one row's contribution)."""
        atol = atol + rms_atol * self.rms(key)
        a = t.detach().to(torch.float64).cpu().numpy() if torch.is_tensor(t) else np.asarray(t, np.float64)
        tag = f"{self.name}:{key} {what}"
        if key + "#full" in self.z:
            ref = self.z[key + "#full"].astype(np.float64)
            assert ref.shape == a.shape, (tag, ref.shape, a.shape)
            err = np.abs(a - ref)
            tol = atol + rtol * np.abs(ref)
            assert np.all(err <= tol), f"{tag}: max err {err.max():.3e} (ref absmax {np.abs(ref).max():.3e})"
            return float(err.max())
        shape = tuple(self.z[key + "#shape"])
        assert shape == a.shape, (tag, shape, a.shape)
        flat = a.reshape(-1)
        for part, mine in (("#head", flat[:N_SAMP]), ("#samp", flat[sample_index(flat.size)])):
            ref = self.z[key + part].astype(np.float64)
            err = np.abs(mine - ref)
            assert np.all(err <= atol + rtol * np.abs(ref)), f"{tag}{part}: max err {err.max():.3e}"
        st = self.z[key + "#stats"]
        mine = np.array([flat.sum(), np.abs(flat).sum(), (flat * flat).sum()])
        # sums of N terms: allow the per-element tolerance accumulated in quadrature-ish fashion
        n = flat.size
        tol = np.array([atol * n ** 0.5 + rtol * st[1], atol * n ** 0.5 + rtol * st[1],
                        2 * atol * st[1] + 2 * rtol * st[2]])
        assert np.all(np.abs(mine - st) <= tol + 1e-12), f"{tag}#stats: {mine} vs {st}"
        return float(np.abs(mine - st).max())


def _rel_l2(self, key, t):
    """||t - ref||_2 / ||ref||_2 over the stored entries of record `key` (the whole tensor, or head + sampled entries of a big one)."""
    a = t.detach().to(torch.float64).cpu().numpy() if torch.is_tensor(t) else np.asarray(t, np.float64)
    if key + "#full" in self.z:
        ref = self.z[key + "#full"].astype(np.float64)
        d = a.reshape(ref.shape) - ref
    else:
        flat = a.reshape(-1)
        ref = np.concatenate([self.z[key + "#head"], self.z[key + "#samp"]]).astype(np.float64)
        d = np.concatenate([flat[:N_SAMP], flat[sample_index(flat.size)]]) - ref
    return float(np.sqrt((d * d).sum()) / (np.sqrt((ref * ref).sum()) + 1e-300))


Golden.rel_l2 = _rel_l2


def case_config(g):
    """(num_class, T, feature_dim, fc_dim, Bs, Bt) of a fixture."""
    return dict(C=int(g.meta("C")), T=int(g.meta("T")), D=int(g.meta("feature_dim")), fc_dim=int(g.meta("fc_dim")),
                Bs=int(g.meta("Bs")), Bt=int(g.meta("Bt")), wseed=int(g.meta("wseed")), wscale=str(g.meta("wscale")),
                xseed=int(g.meta("xseed")), steps=int(g.meta("steps")) if g.has_meta("steps") else 1,
                lr=float(g.meta("lr")) if g.has_meta("lr") else 3e-2,
                clip=float(g.meta("clip")) if g.has_meta("clip") else 20.0,
                short_last=tuple(int(v) for v in g.meta("short_last")) if g.has_meta("short_last") else None,
                agg=str(g.meta("agg")) if g.has_meta("agg") else "trn-m",
                place_adv=tuple(str(v) for v in g.meta("place_adv")) if g.has_meta("place_adv") else None,
                dis_DA=str(g.meta("dis_DA")) if g.has_meta("dis_DA") else "none",
                place_dis=tuple(str(v) for v in g.meta("place_dis")) if g.has_meta("place_dis") else ("N", "Y", "N"),
                alpha=float(g.meta("alpha")) if g.has_meta("alpha") else 0.0,
                ens_DA=str(g.meta("ens_DA")) if g.has_meta("ens_DA") else "none",
                mu=float(g.meta("mu")) if g.has_meta("mu") else 0.0,
                use_bn=str(g.meta("use_bn")) if g.has_meta("use_bn") else "none",
                add_loss_DA=str(g.meta("add_loss_DA")) if g.has_meta("add_loss_DA") else None)


def _has_meta(self, k):
    return ("meta/" + k) in self.z


Golden.has_meta = _has_meta


def step_schedule(cfg):
    """(p, lr_used, batch seeds, n_src, n_tgt) per step, mirroring make_golden.run_case:
    global step s of epoch 1 in a 30-epoch run with `steps` steps per epoch;
    lr for step 0 is lr0, afterwards the DANN value set after the previous step
    (main.py:620-621)."""
    n = cfg["steps"]
    out = []
    lr = cfg["lr"]
    for s in range(n):
        p = float(n + s) / (30 * n)
        ns, nt = cfg["Bs"], cfg["Bt"]
        if cfg["short_last"] is not None and s == n - 1:
            ns, nt = cfg["short_last"]
        out.append(dict(p=p, lr=lr, xseed=cfg["xseed"] + 100 * s, n_src=ns, n_tgt=nt))
        lr = cfg["lr"] / (1.0 + 10 * p) ** 0.75
    return out
