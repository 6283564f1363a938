"""TEST / MEASUREMENT INFRASTRUCTURE - never imported by the product (ta3n_amd/, main.py, train_ddp.py, compat/).

OPT-IN: the reference itself timed on the GPU box's host cores, as a second figure beside bench.py's `cpu_baseline` (whose `value`
is ALWAYS the oracle port, `kind: "port"` - one definition on every box; the reference's figure, when there is one, travels in
`cpu_baseline.reference`).

The reference is pure Python: nothing of it compiles into oracle/_ref/, and a Python reference does not travel with the product.
`__graft_entry__.build()` therefore does NOT stage it.  Whoever wants the reference's own `main.train` timed on a GPU box
runs, explicitly, in a container that has the checkout:

    python -m oracle.reference_runner --stage        # copies the hot path's module files into oracle/_ref/py/ (git-ignored)

`stage()` and `available()` verify every file against the sha256 values PINNED in this file (REFERENCE_SHA256, taken from the
cmhungsteve/TA3N checkout this repository was built against) - a file that differs is neither staged nor ever executed, and a
missing checkout / read-only tree is reported, not raised.  `python -m oracle.reference_runner --config N --threads T --seconds S`
then times, in a process of its own (the import shims patch `torch.Tensor.cuda` to the identity - that must not happen inside a
process that also drives the GPU), the reference's own `main.train()` (main.py:309-667: VideoModel.forward, the loss assembly,
backward, clip_grad_norm_, SGD step, DANN LR) for one batch per call on the same synthetic tensors the GPU path is timed on
(ta3n_amd.synthetic, in memory - TSNDataSet's per-frame file reads are not part of the metric), dropout 0.5 / 0.5, and prints one
JSON line.  Import shims as in tests/golden/ref_shim.py (torchvision arch -> feature dim, colorama, tensorboardX, .cuda() ->
identity, device_count -> 1, accuracy() with .reshape at main.py:820).
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


# sha256 of the reference's files this repository was built and pinned against (cmhungsteve/TA3N as checked out under /root/reference);
# nothing that does not match is copied, imported or executed
REFERENCE_SHA256 = {
    "main.py": "1b7294d3020d36ce6968aae5351ef949e2cc255953b1907f51964646e2cd12f1",
    "models.py": "1d3e87ff49f1a44f6ac6489289c8be5e878e22c53e381b0067a2e8eee5eefc79",
    "TRNmodule.py": "f13c1a50ff6ed7ce9dde1068d451ac03d1e2af8479122195443506aa68109f37",
    "loss.py": "b4bbbeea1dc6c85cbd9aad30b74e240e611951b99efea53898a37064c50924bc",
    "opts.py": "d8206b97209bfd9c2667bb3421fa1e5f53bc59170dd7aa99fa829e170e84c99f",
    "dataset.py": "c66ec054a1b0b8884a0f85243f9e290cbb9cba7c34496a9b03013840fc6a8d66",
    os.path.join("utils", "utils.py"): "4a57be7cb9035b53f429e0c5509970deb92e9702824929b108f37a067063ea66",
}


def _sha256(path: str):
    try:
        with open(path, "rb") as fh:
            return hashlib.sha256(fh.read()).hexdigest()
    except OSError:
        return None


def stage() -> bool:
    """The explicit, opt-in recipe: copy the hot path's module files from the reference checkout into oracle/_ref/py/ (git-ignored)
    - only files whose sha256 equals the pinned one.  Returns False, with the reason on stderr, where there is no checkout, a file
    does not match its pin, or the tree cannot be written; never raises (an existing stage is left alone)."""
    try:
        for f in FILES:
            got = _sha256(os.path.join(REFERENCE, f))
            if got is None:
                print(f"[reference_runner] no reference checkout at {REFERENCE} ({f} missing): nothing staged", file=sys.stderr)
                return False
            if got != REFERENCE_SHA256[f]:
                print(f"[reference_runner] {os.path.join(REFERENCE, f)} does not match the pinned sha256: nothing staged", file=sys.stderr)
                return False
        os.makedirs(os.path.join(STAGE, "utils"), exist_ok=True)
        for f in FILES:
            dst = os.path.join(STAGE, f)
            if os.path.exists(dst):
                os.chmod(dst, 0o644)
            shutil.copyfile(os.path.join(REFERENCE, f), dst)
            os.chmod(dst, 0o444)
        init = os.path.join(STAGE, "utils", "__init__.py")
        if not os.path.exists(init):
            open(init, "w").close()
    except OSError as ex:
        print(f"[reference_runner] staging failed ({ex}): nothing usable staged", file=sys.stderr)
        return False
    return available()


def unstage() -> None:
    shutil.rmtree(os.path.join(HERE, "_ref", "py"), ignore_errors=True)


def available() -> bool:
    """True only when EVERY staged file is present and equals its pinned sha256 - the condition under which this module will
    import and execute them."""
    return all(_sha256(os.path.join(STAGE, f)) == REFERENCE_SHA256[f] for f in FILES)


def staged_sha256() -> dict:
    """{file: first 16 hex digits} of the staged files that match their pins (all of FILES whenever available())."""
    return {f: h[:16] for f, h in REFERENCE_SHA256.items() if _sha256(os.path.join(STAGE, f)) == h}


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


def time_reference(config: int, threads, seconds: float, max_steps: int = 400) -> dict:
    """threads: one count, or several (a bounded probe - two steps each - picks the fastest; every probe time is reported)."""
    import argparse
    import importlib.util
    import io

    import torch
    if not available():
        raise SystemExit("no staged reference under oracle/_ref/py that matches the pinned sha256 values (opt-in: `python -m "
                         "oracle.reference_runner --stage` in a container that has the checkout)")
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
    counts = [int(threads)] if isinstance(threads, int) else [int(t) for t in threads]
    torch.set_num_threads(counts[0])
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
    probe = {}
    with contextlib.redirect_stdout(io.StringIO()):
        step()
        for c in counts:
            torch.set_num_threads(c)
            step()
            t0 = time.perf_counter()
            step()
            step()
            probe[c] = 1e3 * (time.perf_counter() - t0) / 2
        threads = min(probe, key=probe.get)
        torch.set_num_threads(threads)
        step()
        t0 = time.perf_counter()
        n = 0
        while n < max_steps and time.perf_counter() - t0 < seconds:
            step()
            n += 1
        dt = time.perf_counter() - t0
    return {"ms_per_step": 1e3 * dt / n, "steps": n, "threads": threads, "probe_ms_per_step_by_threads": {str(k): round(v, 2) for k, v in probe.items()}, "videos_per_s": (case["Bs"] + case["Bt"]) * n / dt,
            "sha256": staged_sha256(), "torch": torch.__version__,
            "what": "the reference's own main.train (VideoModel.forward, loss assembly, backward, clip_grad_norm_, SGD step) from the staged "
                    "copy of /root/reference, in-memory synthetic features, dropout 0.5 / 0.5, fp32"}


if __name__ == "__main__":
    import argparse as _ap
    p = _ap.ArgumentParser()
    p.add_argument("--stage", action="store_true", help="(opt-in, build container) copy the reference's module files into oracle/_ref/py/ after "
                   "checking them against the pinned sha256 values")
    p.add_argument("--unstage", action="store_true", help="remove oracle/_ref/py/")
    p.add_argument("--config", type=int, default=2)
    p.add_argument("--threads", type=str, default="8", help="torch thread count, or a comma list to probe (the fastest is timed)")
    p.add_argument("--seconds", type=float, default=12.0)
    ns = p.parse_args()
    if ns.unstage:
        unstage()
        print("removed", STAGE)
    elif ns.stage:
        print("staged (every file matches its pinned sha256)" if stage() else "nothing staged", STAGE)
    else:
        print("REFERENCE_JSON " + json.dumps(time_reference(ns.config, [int(t) for t in ns.threads.split(",")], ns.seconds)), flush=True)
