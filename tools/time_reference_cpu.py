"""Build container only (needs /root/reference): times the REFERENCE's own CPU train step - models.VideoModel.forward +
main.train's loss assembly, backward, clip_grad_norm_, SGD step, through the import shims of tests/golden/ref_shim.py - on
in-memory synthetic features (bypassing TSNDataSet), next to the oracle port that bench.py's cpu_baseline times on the GPU
box (where /root/reference does not exist).  Prints one line per configuration for BASELINE.md and writes
profiles/reference_vs_port_cpu.json: the reference-over-port time ratio per BASELINE configuration, which bench.py carries in
`cpu_baseline.reference_over_port` next to the port's number.
usage: python tools/time_reference_cpu.py [threads] [--no-json]"""
import io
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden as mg  # noqa: E402  (installs the shims, imports the reference)
import types  # noqa: E402
# a 1024-d "architecture" for configs[4] (I3D-shaped features): the reference derives feature_dim from torchvision (models.py:125-126),
# which the shim fakes; SURVEY.md 8(d) config 5 prescribes exactly this
sys.modules["torchvision.models"].i3d1024 = lambda pretrained=True: types.SimpleNamespace(fc=types.SimpleNamespace(in_features=1024))
from oracle import ta3n_oracle as orc  # noqa: E402
from ta3n_amd.synthetic import synth_batch, synth_state  # noqa: E402

threads = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else (os.cpu_count() or 1)
torch.set_num_threads(threads)
results = {}
# (bench.py --config number, label, case): configs[4]'s two streams are two such models - the ratio of one stream is the ratio of both
for cnum, name, case in ((2, "config 2/3 (TA3N, 128+74, T=5, 12 classes)", dict(arch="resnet101", fc_dim=512, T=5, C=12, Bs=128, Bt=74, wseed=7, wscale="init", xseed=1)),
                         (1, "config 1 (TemPooling source-only, 128+74, 5 classes)", dict(agg="avgpool", arch="resnet101", fc_dim=512, T=5, C=5, Bs=128, Bt=74, wseed=7, wscale="init", xseed=1)),
                         (4, "config 4 (TA3N, 512+512, T=9, 30 classes)", dict(arch="resnet101", fc_dim=512, T=9, C=30, Bs=512, Bt=512, wseed=7, wscale="init", xseed=1)),
                         (5, "config 5, one stream (TA3N, 128+128, T=12, 1024-d, 12 classes)", dict(arch="i3d1024", fc_dim=512, T=12, C=12, Bs=128, Bt=128, wseed=7, wscale="init", xseed=1))):
    model = mg.build_model(case)
    model.dropout_i.p = model.dropout_v.p = 0.5          # the bench runs with dropout 0.5 / 0.5
    model.dropout_rate_i = model.dropout_rate_v = 0.5
    args = mg.make_args(case)
    mg.ref_main.args = args
    mg.ref_main.gpu_count = 1
    opt = torch.optim.SGD(model.parameters(), args.lr, momentum=0.9, weight_decay=1e-4, nesterov=True)
    crit = torch.nn.CrossEntropyLoss()
    xs, xt, ys, yt = synth_batch(case["C"], case["T"], model.feature_dim, case["Bs"], case["Bt"], seed=1)
    avg = case.get("agg") == "avgpool"
    beta, gamma = ([0.0, 0.0, 0.0], 0.0) if avg else ([0.75, 0.75, 0.5], 0.003)
    wrapped = mg._FakeDP(model)
    log = io.StringIO()

    def ref_step():
        mg.ref_main.train(case["C"], [(xs, ys)], [(xt, yt)], wrapped, crit, crit, opt, 1, log, log, 0, list(beta), gamma, 0)

    n = 10 if case["Bs"] + case["Bt"] < 600 else 4
    for _ in range(2):
        ref_step()
    t0 = time.perf_counter()
    for _ in range(n):
        ref_step()
    ref_ms = 1e3 * (time.perf_counter() - t0) / n
    # the oracle port on the same machine, same threads (what bench.py times on the GPU box)
    cfg = orc.Config(num_class=case["C"], num_segments=case["T"], feature_dim=model.feature_dim, fc_dim=512, frame_aggregation="avgpool" if avg else "trn-m",
                     place_adv=("N", "N", "N") if avg else ("Y", "Y", "Y"), add_loss_DA="none" if avg else "attentive_entropy",
                     use_attn="none" if avg else "TransAttn", compute_dead_branches=True)
    state = orc.TrainState(params=synth_state(orc.param_shapes(cfg), seed=7, scale="init"), lr=3e-2)
    vdim = cfg.feat_dim if avg else 256

    def port_step():
        di = [torch.bernoulli(torch.full((b * case["T"], cfg.feat_dim), 0.5)) / 0.5 for b in (case["Bs"], case["Bt"])]
        dv = [torch.bernoulli(torch.full((b, vdim), 0.5)) / 0.5 for b in (case["Bs"], case["Bt"])]
        orc.train_step(state, xs, xt, ys, beta, gamma, cfg, drop_i=di, drop_v=dv)

    for _ in range(2):
        port_step()
    t0 = time.perf_counter()
    for _ in range(n):
        port_step()
    port_ms = 1e3 * (time.perf_counter() - t0) / n
    cpu = next((ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")), "?")
    print(f"{name}: reference main.train + VideoModel on CPU {ref_ms:.1f} ms/step = {(case['Bs'] + case['Bt']) / ref_ms * 1e3:.0f} videos/s; "
          f"oracle port {port_ms:.1f} ms/step; {threads} torch threads of {os.cpu_count()} vCPU '{cpu}'")
    results[f"configs[{cnum - 1}]"] = {"workload": name, "reference_ms_per_step": round(ref_ms, 2), "port_ms_per_step": round(port_ms, 2),
                                       "reference_over_port": round(ref_ms / port_ms, 4), "steps_timed": n}
if "--no-json" not in sys.argv:
    import hashlib
    out = {"what": "CPU time per train step of the REFERENCE itself (/root/reference main.train + models.VideoModel through the import shims of "
                   "tests/golden/ref_shim.py, in-memory synthetic features, dropout 0.5 / 0.5) and of oracle/ta3n_oracle.py (the port bench.py's "
                   "cpu_baseline times on the GPU box, where the reference does not exist), same process, same torch threads; "
                   "reference_over_port > 1: the reference is slower than the port by that factor",
           "host": {"cpu": cpu, "vcpus": os.cpu_count(), "torch_threads": threads, "torch": torch.__version__},
           "reference_sha256": {f: hashlib.sha256(open(os.path.join("/root/reference", f), "rb").read()).hexdigest()[:16] for f in ("main.py", "models.py", "TRNmodule.py", "loss.py")},
           "oracle_sha256": hashlib.sha256(open(os.path.join(ROOT, "oracle", "ta3n_oracle.py"), "rb").read()).hexdigest()[:16],
           "configs": results}
    with open(os.path.join(ROOT, "profiles", "reference_vs_port_cpu.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote profiles/reference_vs_port_cpu.json")
