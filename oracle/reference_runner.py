"""TEST / MEASUREMENT INFRASTRUCTURE - never imported by the product (ta3n_amd/, main.py, train_ddp.py, compat/).

The REFERENCE ITSELF as the CPU baseline of bench.py (`cpu_baseline.kind == "reference"`).

The reference is pure Python, so nothing of it "compiles into oracle/_ref/"; what can travel to the GPU box is a git-ignored STAGED
copy of the module files the hot path lives in, made from /root/reference by `stage()` below (the recipe; `__graft_entry__.build()`
runs it whenever /root/reference exists, i.e. in the build container) into oracle/_ref/py/ together with their sha256 - oracle/_ref/
is listed in .gitignore and not in .gpurunignore, so it ships with the snapshot like the built .so and never enters history.

`python -m oracle.reference_runner --config N --threads T --seconds S` then times, in a process of its own (the import shims patch
`torch.Tensor.cuda` to the identity - that must not happen inside a process that also drives the GPU), the reference's own
`main.train()` (main.py:309-667: VideoModel.forward, the loss assembly, backward, clip_grad_norm_, SGD step, DANN LR) for one batch
per call on the same synthetic tensors the GPU path is timed on (ta3n_amd.synthetic, in memory - TSNDataSet's per-frame file reads
are not part of the metric), dropout 0.5 / 0.5, and prints one JSON line.  Import shims as in tests/golden/ref_shim.py (torchvision
arch -> feature dim, colorama, tensorboardX, .cuda() -> identity, device_count -> 1, accuracy() with .reshape at main.py:820).
"""
from __future__ import annotations

import hashlib
import json
import os
import shutil
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
STAGE = os.path.join(HERE, "_ref", "py")
REFERENCE = os.environ.get("TA3N_REFERENCE_DIR", "/root/reference")
FILES = ["main.py", "models.py", "TRNmodule.py", "loss.py", "opts.py", "dataset.py", os.path.join("utils", "utils.py")]

# bench.py --config number -> the reference-side case (same shapes as bench.CONFIGS; configs[4] is timed per stream)
CASES = {
    1: dict(agg="avgpool", arch="resnet101", fc_dim=512, T=5, C=5, Bs=128, Bt=74),
    2: dict(agg="trn-m", arch="resnet101", fc_dim=512, T=5, C=12, Bs=128, Bt=74),
    3: dict(agg="trn-m", arch="resnet101", fc_dim=512, T=5, C=12, Bs=128, Bt=74),
    4: dict(agg="trn-m", arch="resnet101", fc_dim=512, T=9, C=30, Bs=512, Bt=512),
    5: dict(agg="trn-m", arch="i3d1024", fc_dim=512, T=12, C=12, Bs=128, Bt=128, streams=2),
}


def stage() -> bool:
    """The recipe: copy the hot path's module files from the reference checkout into oracle/_ref/py/ (git-ignored) and record their
    sha256.  Returns False (and leaves an existing stage alone) where there is no reference checkout - the GPU box."""
    if not os.path.isfile(os.path.join(REFERENCE, "main.py")):
        return False
    os.makedirs(os.path.join(STAGE, "utils"), exist_ok=True)
    sums = {}
    for f in FILES:
        src, dst = os.path.join(REFERENCE, f), os.path.join(STAGE, f)
        if os.path.exists(dst):
            os.chmod(dst, 0o644)
        shutil.copyfile(src, dst)
        os.chmod(dst, 0o444)
        sums[f] = hashlib.sha256(open(dst, "rb").read()).hexdigest()
    init = os.path.join(STAGE, "utils", "__init__.py")
    if not os.path.exists(init):
        open(init, "w").close()
    with open(os.path.join(STAGE, "SHA256SUMS.json"), "w") as fh:
        json.dump(sums, fh, indent=1)
    return True


def available() -> bool:
    return all(os.path.isfile(os.path.join(STAGE, f)) for f in FILES)


def staged_sha256() -> dict:
    try:
        with open(os.path.join(STAGE, "SHA256SUMS.json")) as fh:
            want = json.load(fh)
    except OSError:
        return {}
    return {f: h[:16] for f, h in want.items() if hashlib.sha256(open(os.path.join(STAGE, f), "rb").read()).hexdigest() == h}


def _install_shims():
    import builtins
    import types

    import torch
    builtins.torch = torch                                   # models.py:14 expects `torch` to leak from `from torch.nn.init import *`
    tv, tvm = types.ModuleType("torchvision"), types.ModuleType("torchvision.models")
    for name, dim in dict(resnet18=512, resnet34=512, resnet50=2048, resnet101=2048, resnet152=2048, i3d1024=1024).items():
        setattr(tvm, name, (lambda d: (lambda pretrained=True: types.SimpleNamespace(fc=types.SimpleNamespace(in_features=d))))(dim))
    tv.models = tvm
    sys.modules.update({"torchvision": tv, "torchvision.models": tvm})
    col = types.ModuleType("colorama")
    col.init = lambda **kw: None

    class _Blank:
        def __getattr__(self, k):
            return ""

    col.Fore = col.Back = col.Style = _Blank()
    sys.modules["colorama"] = col
    tbx = types.ModuleType("tensorboardX")
    tbx.SummaryWriter = object
    sys.modules["tensorboardX"] = tbx
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.device_count = lambda: 1                      # main.py:31 (`% gpu_count` at :367-371)


def _fixed_accuracy(output, target, topk=(1,)):
    """main.py:809-822 with .reshape at :820 (the reference line raises on torch >= 1.7 for a non-contiguous slice)."""
    maxk = max(topk)
    _, pred = output.topk(maxk, 1, True, True)
    correct = pred.t().eq(target.view(1, -1).expand_as(pred.t()))
    return [correct[:k].reshape(-1).float().sum(0).mul_(100.0 / target.size(0)) for k in topk]


class _FakeDP:
    """Stands in for nn.DataParallel (main.py:79): main.train uses model.module, model.train(), model(...), model.parameters()."""

    def __init__(self, m):
        self.module = m

    def __call__(self, *a, **k):
        return self.module(*a, **k)

    def train(self, mode=True):
        return self.module.train(mode)

    def parameters(self):
        return self.module.parameters()


def time_reference(config: int, threads: int, seconds: float, max_steps: int = 400) -> dict:
    import argparse
    import importlib.util
    import io

    import torch
    if not available():
        raise SystemExit("no staged reference under oracle/_ref/py (run oracle.reference_runner.stage() in the build container)")
    _install_shims()
    sys.path.insert(0, STAGE)                                # the reference's `import models / loss / opts / dataset / utils.utils`
    sys.path.insert(1, ROOT)
    spec = importlib.util.spec_from_file_location("ta3n_reference_main", os.path.join(STAGE, "main.py"))
    ref_main = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_main)
    import models as ref_models
    assert os.path.realpath(ref_models.__file__).startswith(os.path.realpath(STAGE)), ref_models.__file__
    ref_main.accuracy = _fixed_accuracy
    from ta3n_amd.synthetic import synth_batch, synth_state
    case = CASES[config]
    avg = case["agg"] == "avgpool"
    torch.set_num_threads(threads)
    torch.manual_seed(1)
    m = ref_models.VideoModel(case["C"], "video", case["agg"], "RGB", train_segments=case["T"], val_segments=case["T"], base_model=case["arch"],
                              add_fc=1, fc_dim=case["fc_dim"], dropout_i=0.5, dropout_v=0.5, partial_bn=False, use_bn="none", ens_DA="none",
                              use_attn="none" if avg else "TransAttn", n_attn=1, use_attn_frame="none", verbose=False, share_params="Y")
    sd = m.state_dict()
    sd.update(synth_state({k: tuple(v.shape) for k, v in sd.items()}, seed=7, scale="init"))
    m.load_state_dict(sd)
    a = argparse.Namespace(no_partialbn=True, batch_size=[case["Bs"], case["Bt"], case["Bs"]], baseline_type="video", num_segments=case["T"],
                           pretrain_source=False, pred_normalize="N", tensorboard=False, use_target="none" if avg else "uSv", ens_DA="none",
                           dis_DA="none", adv_DA="none" if avg else "RevGrad", place_adv=["N"] * 3 if avg else ["Y"] * 3,
                           add_loss_DA="none" if avg else "attentive_entropy", use_attn="none" if avg else "TransAttn", clip_gradient=20.0,
                           verbose=False, print_freq=1, show_freq=10 ** 9, lr_adaptive="dann", lr=3e-2, save_attention=-1, epochs=30, add_fc=1,
                           momentum=0.9, weight_decay=1e-4, place_dis=["N", "Y", "N"])
    ref_main.args = a
    ref_main.gpu_count = 1
    opt = torch.optim.SGD(m.parameters(), a.lr, momentum=0.9, weight_decay=1e-4, nesterov=True)
    crit = torch.nn.CrossEntropyLoss()
    xs, xt, ys, yt = synth_batch(case["C"], case["T"], m.feature_dim, case["Bs"], case["Bt"], seed=1234)
    beta, gamma = ([0.0, 0.0, 0.0], 0.0) if avg else ([0.75, 0.75, 0.5], 0.003)
    wrapped, log = _FakeDP(m), io.StringIO()
    streams = case.get("streams", 1)

    def step():
        for _ in range(streams):                             # (configs[4]: two such models per step; one model stepped twice costs the same)
            ref_main.train(case["C"], [(xs, ys)], [(xt, yt)], wrapped, crit, crit, opt, 1, log, log, 0, list(beta), gamma, 0)

    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        step()
        step()
        t0 = time.perf_counter()
        n = 0
        while n < max_steps and time.perf_counter() - t0 < seconds:
            step()
            n += 1
        dt = time.perf_counter() - t0
    return {"ms_per_step": 1e3 * dt / n, "steps": n, "threads": threads, "videos_per_s": (case["Bs"] + case["Bt"]) * n / dt,
            "sha256": staged_sha256(), "torch": torch.__version__,
            "what": "the reference's own main.train (VideoModel.forward, loss assembly, backward, clip_grad_norm_, SGD step) from the staged "
                    "copy of /root/reference, in-memory synthetic features, dropout 0.5 / 0.5, fp32"}


if __name__ == "__main__":
    import argparse as _ap
    p = _ap.ArgumentParser()
    p.add_argument("--stage", action="store_true", help="(build container) copy the reference's module files into oracle/_ref/py/")
    p.add_argument("--config", type=int, default=2)
    p.add_argument("--threads", type=int, default=8)
    p.add_argument("--seconds", type=float, default=12.0)
    ns = p.parse_args()
    if ns.stage:
        print("staged" if stage() else "no reference checkout here", STAGE)
    else:
        print("REFERENCE_JSON " + json.dumps(time_reference(ns.config, ns.threads, ns.seconds)), flush=True)
