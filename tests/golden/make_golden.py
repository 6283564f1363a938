#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REFERENCE ITSELF on CPU.

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden.py
The fixtures pin oracle/ta3n_oracle.py (tests/test_oracle_golden.py) and, through
it and directly, the HIP path (tests/test_gpu_parity.py).  Everything that
produces numbers here is reference code: ``models.VideoModel.forward`` and
``main.train`` (loss assembly, backward, clip_grad_norm_, SGD step, DANN LR),
imported unmodified through ref_shim.  Weights and inputs come from
ta3n_amd.synthetic (numpy Generator) so no large tensors need to be stored.
"""
import argparse
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_shim  # noqa: E402

ref_shim.install()
import importlib.util  # noqa: E402

# the reference's main.py BY PATH: this repository has a main.py of its own at its root, which `import main` would find first
_spec = importlib.util.spec_from_file_location("ta3n_reference_main", os.path.join(ref_shim.REF, "main.py"))
ref_main = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(ref_main)  # (pulls the reference's models / loss / opts through sys.path, set up by ref_shim)
assert "dis_DA != 'none'" in open(ref_main.__file__).read() and ref_main.__file__.startswith(ref_shim.REF)
from models import VideoModel as RefVideoModel  # noqa: E402

from ta3n_amd.synthetic import synth_batch, synth_state  # noqa: E402

ref_main.accuracy = ref_shim.fixed_accuracy

N_SAMP = 64


def sample_index(numel):
    return (np.arange(N_SAMP, dtype=np.int64) * 7919 + 13) % numel


def put(store, key, t):
    a = t.detach().to(torch.float64).cpu().numpy() if torch.is_tensor(t) else np.asarray(t, dtype=np.float64)
    if a.size <= 8192:
        store[key + "#full"] = a.astype(np.float32) if a.dtype != np.int64 else a
    else:
        flat = a.reshape(-1)
        store[key + "#stats"] = np.array([flat.sum(), np.abs(flat).sum(), (flat * flat).sum()], dtype=np.float64)
        store[key + "#head"] = flat[:N_SAMP].astype(np.float32)
        store[key + "#samp"] = flat[sample_index(flat.size)].astype(np.float32)
        store[key + "#shape"] = np.array(a.shape, dtype=np.int64)


class _FakeDP:
    """Stands in for nn.DataParallel (main.py:79): main.train uses model.module,
    model.train(), model(...), model.parameters()."""

    def __init__(self, m):
        self.module = m

    def __call__(self, *a, **k):
        return self.module(*a, **k)

    def train(self, mode=True):
        return self.module.train(mode)

    def parameters(self):
        return self.module.parameters()


def build_model(case):
    torch.manual_seed(1)
    avg = case.get("agg", "trn-m") == "avgpool"       # BASELINE configs[0]: TemPooling, source-only (script_train_val.sh:103-119)
    m = RefVideoModel(case["C"], "video", "avgpool" if avg else "trn-m", "RGB", train_segments=case["T"], val_segments=case["T"],
                      base_model=case["arch"], add_fc=1, fc_dim=case["fc_dim"], dropout_i=0.0, dropout_v=0.0,
                      partial_bn=False, use_bn=case.get("use_bn", "none"), ens_DA=case.get("ens_DA", "none"), use_attn="none" if avg else "TransAttn", n_attn=1,
                      use_attn_frame="none", verbose=False, share_params="Y")
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = m.state_dict()
    sd.update(synth_state(shapes, seed=case["wseed"], scale=case["wscale"]))
    m.load_state_dict(sd)
    return m


def make_args(case):
    a = argparse.Namespace()
    a.no_partialbn = True
    a.batch_size = [case["Bs"], case["Bt"], case["Bs"]]
    a.baseline_type = "video"
    a.num_segments = case["T"]
    a.pretrain_source = False
    a.pred_normalize = "N"
    a.tensorboard = False
    a.use_target = "uSv"
    a.ens_DA = "none"
    a.dis_DA = "none"
    a.adv_DA = "RevGrad"
    a.place_adv = ["Y", "Y", "Y"]
    a.add_loss_DA = "attentive_entropy"
    a.use_attn = "TransAttn"
    a.clip_gradient = case.get("clip", 20.0)
    a.verbose = False
    a.print_freq = 1
    a.show_freq = 10 ** 9
    a.lr_adaptive = "dann"
    a.lr = case.get("lr", 3e-2)
    a.save_attention = -1
    a.epochs = 30
    a.add_fc = 1
    a.momentum = 0.9
    a.weight_decay = 1e-4
    if case.get("agg", "trn-m") == "avgpool":      # use_target none switches every DA option off (script_train_val.sh:103-119)
        a.use_target = "none"
        a.adv_DA = "none"
        a.place_adv = ["N", "N", "N"]
        a.add_loss_DA = "none"
        a.use_attn = "none"
        if case.get("place_adv"):                  # TemPooling + RevGrad (the DA rows of the paper's TemPooling table): no attention
            a.use_target = "uSv"
            a.adv_DA = "RevGrad"
            a.place_adv = list(case["place_adv"])
    # discrepancy-based DA (main.py:452-505) and MCD (main.py:402, 447, 548-556): SURVEY 8f rank 4
    a.dis_DA = case.get("dis_DA", "none")
    a.place_dis = list(case.get("place_dis", ("N", "Y", "N")))      # script_train_val.sh:148
    a.ens_DA = case.get("ens_DA", "none")
    if "add_loss_DA" in case:                      # (e.g. MCD without the attentive-entropy term: main.py:559-562 off)
        a.add_loss_DA = case["add_loss_DA"]
    return a


def run_case(name, case):
    store = {}
    meta = dict(case)
    model = build_model(case)
    D = model.feature_dim
    T, C = case["T"], case["C"]
    avg = case.get("agg", "trn-m") == "avgpool"
    avg_da = avg and bool(case.get("place_adv"))
    beta = [0.0, 0.0, 0.0] if (avg and not avg_da) else [0.75, 0.75, 0.5]
    gamma = 0.0 if avg else case.get("gamma", 0.003)

    # ---- (1) plain forward through the reference model (train mode, dropout 0) ----
    xs, xt, ys, yt = synth_batch(C, T, D, case["Bs"], case["Bt"], seed=case["xseed"])
    model.train()
    with torch.no_grad():
        out = model(xs, xt, beta, 0, True, False)
    attn_s, out_s, out_s2, pd_s, feat_s, attn_t, out_t, out_t2, pd_t, feat_t = out
    put(store, "fwd/attn_s", attn_s); put(store, "fwd/attn_t", attn_t)
    put(store, "fwd/out_s", out_s); put(store, "fwd/out_t", out_t)
    if not avg or avg_da:      # (avgpool source-only: the discriminators are forwarded but feed nothing)
        for i, nm in enumerate(("rel", "vid", "frm")):       # avgpool: "rel" is the video logits once more (models.py:707-708)
            put(store, f"fwd/pd_s_{nm}", pd_s[i]); put(store, f"fwd/pd_t_{nm}", pd_t[i])
    for i, nm in enumerate(("y", "v", "f1")):
        put(store, f"fwd/feat_s_{nm}", feat_s[i]); put(store, f"fwd/feat_t_{nm}", feat_t[i])
    if case.get("ens_DA", "none") == "MCD":
        put(store, "fwd/out_s2", out_s2); put(store, "fwd/out_t2", out_t2)

    # ---- (2) the reference's own train loop for n_steps steps ----
    args = make_args(case)
    ref_main.args = args
    ref_main.gpu_count = 1
    opt = torch.optim.SGD(model.parameters(), args.lr, momentum=args.momentum,
                          weight_decay=args.weight_decay, nesterov=True)
    crit = torch.nn.CrossEntropyLoss()
    crit_d = torch.nn.CrossEntropyLoss()
    n_steps = case.get("steps", 1)
    short = case.get("short_last", None)     # (n_src, n_tgt) for the last step: exercises dummy padding
    src_batches, tgt_batches = [], []
    for s in range(n_steps):
        bxs, bxt, bys, byt = synth_batch(C, T, D, case["Bs"], case["Bt"], seed=case["xseed"] + 100 * s)
        if short is not None and s == n_steps - 1:
            bxs, bys, bxt, byt = bxs[:short[0]], bys[:short[0]], bxt[:short[1]], byt[:short[1]]
        src_batches.append((bxs, bys)); tgt_batches.append((bxt, byt))
    log, log_short = io.StringIO(), io.StringIO()
    lrs = []
    wrapped = _FakeDP(model)
    # main.train runs one whole epoch; feed one step at a time so grads / params /
    # lr can be captured per step.  start_steps = epoch * len(loader) (main.py:334),
    # so the loader handed in is padded in front with None and sliced by a proxy.

    for s in range(n_steps):
        # one-batch loaders: i == 0 inside train(); to reproduce global step s of an
        # n_steps-long epoch 1 we need p = (s + 1*n_steps) / (epochs*n_steps).
        # With len(loader) == 1: p' = (0 + epoch') / epochs'.  Choose epochs' =
        # epochs*n_steps and epoch' = n_steps + s.
        args.epochs = 30 * n_steps
        epoch_eff = n_steps + s
        ref_main.train(C, [src_batches[s]], [tgt_batches[s]], wrapped, crit, crit_d, opt, epoch_eff,
                       log, log_short, case.get("alpha", 0), list(beta), gamma, case.get("mu", 0))
        lrs.append(opt.param_groups[0]["lr"])
        p = float(epoch_eff) / args.epochs
        store[f"step{s}/p"] = np.array([p])
        store[f"step{s}/lr_after"] = np.array([opt.param_groups[0]["lr"]])
        for k, v in model.named_parameters():
            if v.grad is not None:
                put(store, f"step{s}/clipped_grad/{k}", v.grad)
            put(store, f"step{s}/param/{k}", v)
    if case.get("use_bn", "none") != "none":
        # AdaBN / AutoDIAL (models.py:490-543, 569-570): the running statistics after the train-mode passes above (the plain
        # forward of (1) counts as one: BatchNorm updates its buffers whenever it runs in train mode) and an eval-mode forward
        # through them (main.validate's call, main.py:707)
        for k, v in model.state_dict().items():
            if "bn_shared" in k:
                put(store, f"final/state/{k}", v.double() if v.dtype != torch.float32 else v)
        model.eval()
        with torch.no_grad():
            ev = model(xs, xs, [0, 0, 0], 0, False, False)
        put(store, "eval/out_t", ev[6]); put(store, "eval/feat_t_v", ev[9][1])
        model.train()
    meta["live"] = [k for k, v in model.named_parameters() if v.grad is not None]
    meta["log"] = log.getvalue()
    store["meta/live"] = np.array(meta["live"])
    store["meta/log"] = np.array(meta["log"])
    for k, v in case.items():
        if isinstance(v, (int, float)):
            store[f"meta/{k}"] = np.array([v])
        elif isinstance(v, str):
            store[f"meta/{k}"] = np.array(v)
        elif isinstance(v, (tuple, list)):
            store[f"meta/{k}"] = np.array(v)
    store["meta/feature_dim"] = np.array([D])
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **store)
    print(name, "->", path, os.path.getsize(path) // 1024, "KiB", "lrs", lrs)
    print(log.getvalue().strip().splitlines()[-1][:200])


CASES = {
    # small shapes the oracle and a debug GPU run finish instantly
    "tiny_T5": dict(arch="resnet18", fc_dim=64, T=5, C=12, Bs=6, Bt=4, wseed=7, wscale="trained", xseed=1234,
                    steps=3, short_last=(5, 3), lr=2e-3),
    "tiny_T3": dict(arch="resnet18", fc_dim=32, T=3, C=5, Bs=4, Bt=5, wseed=8, wscale="trained", xseed=99, steps=1, lr=2e-3),
    "tiny_T9": dict(arch="resnet18", fc_dim=32, T=9, C=30, Bs=5, Bt=5, wseed=9, wscale="trained", xseed=77, steps=2, lr=2e-3),
    "tiny_T2": dict(arch="resnet18", fc_dim=32, T=2, C=5, Bs=3, Bt=2, wseed=10, wscale="trained", xseed=5, steps=1, lr=2e-3),
    # clip active: tiny max-norm so the clip branch of clip_grad_norm_ is exercised
    "tiny_clip": dict(arch="resnet18", fc_dim=64, T=5, C=12, Bs=6, Bt=4, wseed=11, wscale="trained", xseed=4321,
                      steps=2, clip=0.05, lr=2e-3),
    # BASELINE config 2/3 shape, trained-scale and reference-init weights
    "headline": dict(arch="resnet101", fc_dim=512, T=5, C=12, Bs=128, Bt=74, wseed=7, wscale="trained",
                     xseed=1234, steps=2, lr=2e-3),
    "headline_init": dict(arch="resnet101", fc_dim=512, T=5, C=12, Bs=128, Bt=74, wseed=7, wscale="init",
                          xseed=1234, steps=1),
    # config-5-like shape: T=12, single 1024-d stream is not expressible (no 1024-d arch); 2048-d used
    # BASELINE configs[0] (hmdb_ucf_small, TemPooling, source-only): avgpool aggregation, every DA option off
    "tiny_avgpool": dict(agg="avgpool", arch="resnet18", fc_dim=64, T=5, C=5, Bs=6, Bt=4, wseed=13, wscale="trained", xseed=55,
                         steps=3, short_last=(5, 3), lr=2e-3),
    "config1_avgpool": dict(agg="avgpool", arch="resnet101", fc_dim=512, T=5, C=5, Bs=128, Bt=74, wseed=14, wscale="trained",
                            xseed=66, steps=2, lr=2e-3),
    "mid_T12": dict(arch="resnet101", fc_dim=128, T=12, C=12, Bs=16, Bt=16, wseed=12, wscale="trained",
                    xseed=31, steps=1, lr=2e-3),
    # TemPooling + RevGrad: avgpool with the video- and frame-level adversarial branches; with
    # place_adv[0] = 'Y' the reference counts the video-level loss twice (its relation slot holds the video logits)
    "tiny_avgpool_da": dict(agg="avgpool", place_adv=("N", "Y", "Y"), arch="resnet18", fc_dim=64, T=5, C=5, Bs=6, Bt=4, wseed=15,
                            wscale="trained", xseed=57, steps=3, short_last=(5, 3), lr=2e-3),
    "tiny_avgpool_da3": dict(agg="avgpool", place_adv=("Y", "Y", "Y"), arch="resnet18", fc_dim=64, T=3, C=7, Bs=4, Bt=5, wseed=16,
                             wscale="trained", xseed=58, steps=2, lr=2e-3),
    "tiny_avgpool_dav": dict(agg="avgpool", place_adv=("N", "Y", "N"), arch="resnet18", fc_dim=32, T=4, C=5, Bs=5, Bt=3, wseed=17,
                             wscale="trained", xseed=59, steps=2, lr=2e-3),
    "tempooling_da": dict(agg="avgpool", place_adv=("N", "Y", "Y"), arch="resnet101", fc_dim=512, T=5, C=12, Bs=128, Bt=74, wseed=18,
                          wscale="trained", xseed=60, steps=2, lr=2e-3),
    # discrepancy-based DA on top of TA3N's adversarial branches (dis_DA, main.py:452-505; loss.py:46-120) and MCD
    # (ens_DA, second classifier + a second, gradient-reversed forward: models.py:276-279, 682-684, 716-720; main.py:548-556)
    "tiny_dan": dict(arch="resnet18", fc_dim=64, T=5, C=12, Bs=6, Bt=4, wseed=21, wscale="trained", xseed=201, steps=2, lr=2e-3,
                     dis_DA="DAN", place_dis=("N", "Y", "N"), alpha=1.0),
    "tiny_dan_all": dict(arch="resnet18", fc_dim=32, T=3, C=5, Bs=4, Bt=5, wseed=22, wscale="trained", xseed=202, steps=2, lr=2e-3,
                         dis_DA="DAN", place_dis=("Y", "Y", "N"), alpha=0.5),   # (place_dis[2] = 'Y' raises in the reference: loss.py:49 on 3-D frame features)
    "tiny_jan": dict(arch="resnet18", fc_dim=64, T=5, C=12, Bs=6, Bt=4, wseed=23, wscale="trained", xseed=203, steps=2, lr=2e-3,
                     dis_DA="JAN", alpha=1.0),
    "tiny_mcd": dict(arch="resnet18", fc_dim=64, T=5, C=12, Bs=6, Bt=4, wseed=24, wscale="trained", xseed=204, steps=2, lr=2e-3,
                     ens_DA="MCD", mu=0.5),
    # use_bn AdaBN / AutoDIAL (models.py:195-198, 490-543, 569-570): domain-specific BatchNorm between the shared FC and its ReLU
    "tiny_adabn": dict(arch="resnet18", fc_dim=64, T=5, C=12, Bs=6, Bt=4, wseed=26, wscale="trained", xseed=206, steps=3,
                       short_last=(5, 3), lr=2e-3, use_bn="AdaBN"),
    "tiny_autodial": dict(arch="resnet18", fc_dim=32, T=3, C=5, Bs=4, Bt=5, wseed=27, wscale="trained", xseed=207, steps=2, lr=2e-3,
                          use_bn="AutoDIAL"),
    "mid_adabn": dict(arch="resnet101", fc_dim=128, T=5, C=12, Bs=16, Bt=12, wseed=28, wscale="trained", xseed=208, steps=2,
                      lr=2e-3, use_bn="AdaBN"),
    # the same options on TemPooling (the TemPooling + X rows of the paper's tables): avgpool aggregation, RevGrad on the video and
    # frame level (place_adv N Y Y), no attention
    "tiny_avgpool_dan_mcd": dict(agg="avgpool", place_adv=("N", "Y", "Y"), arch="resnet18", fc_dim=64, T=5, C=5, Bs=6, Bt=4, wseed=31,
                                 wscale="trained", xseed=301, steps=2, lr=2e-3, dis_DA="DAN", place_dis=("Y", "Y", "N"), alpha=1.0,
                                 ens_DA="MCD", mu=0.5),
    "tiny_avgpool_jan": dict(agg="avgpool", place_adv=("N", "Y", "N"), arch="resnet18", fc_dim=32, T=4, C=5, Bs=5, Bt=3, wseed=32,
                             wscale="trained", xseed=302, steps=2, lr=2e-3, dis_DA="JAN", alpha=0.5),
    "tiny_avgpool_adabn": dict(agg="avgpool", place_adv=("N", "Y", "Y"), arch="resnet18", fc_dim=64, T=5, C=5, Bs=6, Bt=4, wseed=33,
                               wscale="trained", xseed=303, steps=3, short_last=(5, 3), lr=2e-3, use_bn="AdaBN"),
    # MCD WITHOUT attentive entropy, four steps: nothing but `loss` keeps an iteration's graph alive, so the previous iteration's
    # autograd node dies in the middle of the next one (workspace pool handed out a buffer that was in use from step 2 on)
    "tiny_mcd_noent": dict(arch="resnet18", fc_dim=64, T=5, C=12, Bs=6, Bt=4, wseed=29, wscale="trained", xseed=209, steps=4, lr=2e-3,
                           ens_DA="MCD", mu=0.5, add_loss_DA="none"),
    "tiny_avgpool_mcd_noent": dict(agg="avgpool", place_adv=("N", "Y", "Y"), arch="resnet18", fc_dim=64, T=5, C=5, Bs=6, Bt=4, wseed=34,
                                   wscale="trained", xseed=304, steps=4, lr=2e-3, ens_DA="MCD", mu=0.5),
    "mid_dan_mcd": dict(arch="resnet101", fc_dim=128, T=5, C=12, Bs=16, Bt=12, wseed=25, wscale="trained", xseed=205, steps=2,
                        lr=2e-3, dis_DA="DAN", place_dis=("Y", "Y", "N"), alpha=1.0, ens_DA="MCD", mu=1.0),
}

if __name__ == "__main__":
    which = sys.argv[1:] or list(CASES)
    for nm in which:
        run_case(nm, CASES[nm])
