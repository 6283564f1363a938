#!/usr/bin/env python
"""Benchmark of the TA3N temporal-adversarial train step on MI355X.

Metric (BASELINE.json): src+tgt videos/sec per train step, UCF->HMDB_full 5-seg
TA3N (trn-m, RevGrad x3, TransAttn, attentive entropy; 128 source + 74 target
videos per GPU and step, 2048-d features, 12 classes, dropout 0.5/0.5).  --config N
times another BASELINE configuration (1: TemPooling source-only, 3: the headline in
fp32, 4: 30 classes / 9 segments / 512+512 videos, 5: two-stream 1024-d / 12 segments);
`config.workload` names it.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python bench.py --gpus N --steps K --warmup W          # N > 1, bare: starts its own N ranks (launch_ranks), fails if the box has < N GPUs
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W      # the same job under a caller's launcher (WORLD_SIZE must equal N)

One "step" = forward + loss + backward + (RCCL all-reduce of the flat gradient
buffer when N > 1) + clip + Nesterov SGD on synthetic features already resident
in HBM.  Default arithmetic: BASELINE configs[1] (bf16 MFMA operands, fp32
accumulation and fp32 state); --dtype f32 is configs[2]'s.  At N = 1 the other
arithmetic is timed in the same process and reported under "other_arithmetic" (at every N:
with --gpus 8 that is configs[2], fp32 DDP).  Weak scaling: every rank processes its own 128+74 videos.  Rank 0
prints ONE JSON line; `roofline` is measured live with HIP events on the launch
stream, `cpu_baseline` times the reference's own main.train (a staged, git-ignored copy
of its module files, oracle/reference_runner.py) and the CPU oracle (the port)
on a bounded sample on this host's cores.
"""
import argparse
import gc
import json
import math
import os
import sys
import time

# dmabuf IPC for cross-process device memory (RCCL ranks, the opt-in peer transport): must be in the environment before the HIP runtime
# initialises, i.e. before `import torch` (only tests/conftest.py used to set it)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from ta3n_amd.engine import TrainEngine, beta_dann, lr_dann  # noqa: E402
from ta3n_amd.synthetic import synth_batch, synth_state  # noqa: E402

# BASELINE config 2/3 (script_train_val.sh: bS=128, bS_2=128*840/1438=74, 5 segments, fc_dim 512, 12 classes)
CFG = dict(Bs=128, Bt=74, T=5, D=2048, F=512, C=12, NB=256)
# --config N (SURVEY.md 8d numbering = BASELINE.json configs[N-1]); 2 with --dtype bf16 is the headline, 3 is the same shape in fp32
CONFIGS = {
    1: dict(shape=dict(Bs=128, Bt=74, T=5, D=2048, F=512, C=5, NB=256), agg="avgpool", streams=1, dtype="f32",
            name="hmdb_ucf_small shape, TemPooling (avgpool), source-only, 128 src + 74 tgt videos (target forwarded), 5 classes"),
    2: dict(shape=CFG, agg="trn-m", streams=1, dtype="bf16",
            name="UCF->HMDB_full TA3N train step: trn-m 5 segments, RevGrad x3, TransAttn, attentive entropy, 128 src + 74 tgt videos per "
                 "GPU-step, 2048-d features, 12 classes"),
    3: dict(shape=CFG, agg="trn-m", streams=1, dtype="f32", name="UCF->HMDB_full TA3N (same as config 2), fp32"),
    4: dict(shape=dict(Bs=512, Bt=512, T=9, D=2048, F=512, C=30, NB=256), agg="trn-m", streams=1, dtype="bf16",
            name="synthetic Kinetics->Gameplay shape: TA3N, 30 classes, 9 segments, 512 src + 512 tgt videos per GPU-step, 2048-d features"),
    5: dict(shape=dict(Bs=128, Bt=128, T=12, D=1024, F=512, C=12, NB=256), agg="trn-m", streams=2, dtype="bf16",
            name="two-stream RGB+Flow: two TA3N models on 1024-d I3D-shaped features, 12 segments, attentive entropy on, 128 src + 128 tgt "
                 "videos per stream and GPU-step, class logits summed (ta3n_amd/two_stream.py)"),
}
PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2516.6    # same guide: v_mfma_f32_32x32x16_bf16, dense (16 x the fp32 rate)
HBM_PEAK_GBS = 8000.0
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "gemm_traffic.json")   # written by tools/measure_traffic.py from rocprofv3 PMC passes


def measured_traffic(dtype, key=None):
    """HBM-side bytes per GEMM launch (FETCH_SIZE x 2 [gfx950 half-count correction] + WRITE_SIZE, average over the GEMM
    launches of a fused step) of the headline configuration, as measured by tools/measure_traffic.py on a GPU box and
    committed under profiles/ together with the raw counter tables.  PMC collection needs its own rocprofv3 passes (it
    cannot run inside the timed loop), so the line carries the number of the LAST measurement plus the hash of the kernel
    sources it was taken on and whether that is still the tree's hash: a stale number is visible as such."""
    try:
        with open(TRAFFIC_FILE) as f:
            t = json.load(f)
        from ta3n_amd.build import source_hash
        e = t[key or dtype]
        return e["bytes_per_gemm_launch"], {"file": "profiles/gemm_traffic.json", "measured_on_sources": t.get("source_hash"),
                                            "current_sources": source_hash(), "fresh": t.get("source_hash") == source_hash(),
                                            "passes": e.get("passes")}
    except Exception as ex:      # no measurement committed for this arithmetic
        return None, {"file": None, "error": str(ex)[:80]}


def gemm_launch_table(eng, phases, bf16, split):
    """Per GEMM launch of the fused step, timed ALONE (ta3n_time_phases: HIP events on the launch stream, no optimiser riders):
    [{tile, workgroups, us, gflop, tflops, frac_of_mfma_peak}] - gflop is what the launch computes as the plan tiled it (valid rows x
    columns x K of every tile, from the plan description), the peak the dense MFMA peak of the arithmetic (bf16 2.5 PF, a third of it
    for the three-product split, fp32 157.3 TF)."""
    desc = [ph for ph in eng.plan.description["phases"] if ph["kind"] == 0 and ph["group"] == 4]
    timed = [p for p in phases if p[0] == 0]
    peak = (PEAK_BF16_MFMA_TFLOPS / 3 if split else PEAK_BF16_MFMA_TFLOPS) if (bf16 or split) else PEAK_FP32_MFMA_TFLOPS
    out = []
    for ph, t in zip(desc, timed):
        us = 1e3 * t[3]
        tf = ph.get("flops", 0.0) / max(us * 1e-6, 1e-12) / 1e12
        out.append({"tile": ph["tile"] % 1000 + 1000 * ((ph["tile"] // 1000) & 15), "blocks_per_wave": [ph.get("rm", 1), ph.get("rn", 1)],
                    "workgroups": ph["task_count"], "us": round(us, 2), "gflop": round(ph.get("flops", 0.0) / 1e9, 3),
                    "tflops": round(tf, 1), "frac_of_mfma_peak": round(tf / peak, 4)})
    return out


def algorithmic_gemm_flops(Bs, Bt, T, D, F, C, NB):
    """SURVEY.md 8(d) without the dead frame classifier (its output feeds nothing
    for baseline_type='video'): fwd + dgrad + wgrad, no dgrad for the first layer."""
    B = Bs + Bt
    from ta3n_amd import _lib
    gathered = sum(len(t) for scale in _lib.relation_table(T) for t in scale)   # 32 frame rows / video at T=5
    fwd = (2 * B * T * D * F + 2 * B * T * F * (F + 2) + 2 * B * NB * F * gathered +
           (T - 1) * 2 * B * NB * (NB + 2) + 2 * B * NB * C + 2 * B * NB * (NB + 2))
    return 3 * fwd - 2 * B * T * D * F


def algorithmic_gemm_bytes_bf16(Bs, Bt, T, D, F, C, NB, agg="trn-m"):
    """SURVEY.md 8(d) bytes of the contraction launches with bf16 operands: the input once (2 B/element), every live
    weight read twice as bf16 (forward + backward) and its fp32 gradient written once; activations are assumed to stay on
    chip.  (The optimiser's 5 x 4 B/parameter belong to the SGD kernel, not to this one.)"""
    from ta3n_amd import _lib
    B = Bs + Bt
    plan = _lib.Plan(Bs, Bt, T, D, F, C, 0x1F if agg == "trn-m" else 0, aggregation=_lib.AGG_TRN_M if agg == "trn-m" else _lib.AGG_AVGPOOL)
    live = sum(int(torch.tensor(shape).prod()) for _, _, shape, lv in plan.params if lv)
    return B * T * D * 2 + live * (2 + 2 + 4)




def algorithmic_flops(cfg):
    """GEMM FLOPs of one train step of a CONFIGS entry (per stream)."""
    sh = cfg["shape"]
    if cfg["agg"] == "avgpool":      # shared FC (fwd + wgrad, no input gradient) + classifier (fwd, dgrad, wgrad)
        B = sh["Bs"] + sh["Bt"]
        return 2 * (2 * B * sh["T"] * sh["D"] * sh["F"]) + 3 * (2 * B * sh["F"] * sh["C"])
    return algorithmic_gemm_flops(**sh)


HBM_ACHIEVABLE_GBS = 6300.0       # MI355X_MICROARCH.md: 6.29 TB/s measured (float4 copy); SURVEY.md 8(d)'s single-GPU bounds use it


def whole_step_bound(conf, dtype, eng):
    """t >= max(algorithmic FLOPs / MFMA peak of the arithmetic, algorithmic bytes / achievable HBM bandwidth) for ONE
    train step of a CONFIGS entry (all its streams).  Bytes: the input once (fp32 features, the survey's convention) + 8 x 4 B per
    live parameter (weights read in forward and backward, gradient written, optimiser: read p, g, m, write p, m)."""
    sh = conf["shape"]
    live = sum(math.prod(s_) for _, _, s_, lv in eng.plan.params if lv)
    nbytes = (sh["Bs"] + sh["Bt"]) * sh["T"] * sh["D"] * 4 + live * 4 * 8
    flops = algorithmic_flops(conf)
    peak = PEAK_BF16_MFMA_TFLOPS if dtype == "bf16" else PEAK_BF16_MFMA_TFLOPS / 3 if dtype == "f32x3" else PEAK_FP32_MFMA_TFLOPS
    t_flop, t_bytes = flops / (peak * 1e12) * 1e6, nbytes / (HBM_ACHIEVABLE_GBS * 1e9) * 1e6
    n = conf["streams"]
    return {"bound": "mfma" if t_flop >= t_bytes else "hbm", "t_flop_us": n * t_flop, "t_bytes_us": n * t_bytes,
            "bound_us": n * max(t_flop, t_bytes), "flops": n * flops, "bytes": n * nbytes}


# ---- multi-GPU projection: a stated prediction for the 8-GPU run to test -------------------------------------
XGMI_LINK_GBS = 153.0      # MI355X_MICROARCH.md / SURVEY.md 5: 7 links x ~153 GB/s per GPU, full mesh (one link per peer)


def scaling_projection(step_ms_by_config):
    """Weak-scaling efficiency t_1 / (t_1 + t_norm + t_AR(N)) per BASELINE configuration: one sum all-reduce of the flat live-gradient
    prefix per step, exposed in full (clip_grad_norm_ needs the norm of the SUMMED gradient before the first update),
    plus the separate gradient-norm pass over the reduced buffer.  t_AR is bracketed by two models of RCCL on a full xGMI mesh:
    `ring` - 2 (N-1)/N S bytes through ONE link per direction at 60-70 % of its 153 GB/s + 2 (N-1) hops of ~4 us;
    `direct` - reduce-scatter + all-gather with every peer at once: 2 S/N bytes per link + 2 x ~12 us.
    step_ms_by_config: {config number: measured single-GPU ms per step in THIS run} (entries without a measurement are left out)."""
    from ta3n_amd import _lib
    out = {}
    for cnum, ms in sorted(step_ms_by_config.items()):
        cf = CONFIGS[cnum]
        sh = cf["shape"]
        trn = cf["agg"] == "trn-m"
        plan = _lib.Plan(sh["Bs"], sh["Bt"], sh["T"], sh["D"], sh["F"], sh["C"], 0x1F if trn else 0,
                         aggregation=_lib.AGG_TRN_M if trn else _lib.AGG_AVGPOOL)
        live = sum(math.prod(s_) for _, _, s_, lv in plan.params if lv) * cf["streams"]
        row = {"workload": f"configs[{cnum - 1}]", "step_ms_1gpu": round(ms, 4), "message_bytes_fp32": 4 * live, "videos_per_gpu_step": sh["Bs"] + sh["Bt"]}
        t_norm = 4 * live / (HBM_ACHIEVABLE_GBS * 1e9) * 1e3 + 0.003          # one read of the reduced buffer + a launch
        for transport, esz in (("fp32", 4), ("bf16", 2)):
            S = esz * live
            for N in (2, 4, 8):
                ring = 2 * (N - 1) / N * S / (0.65 * XGMI_LINK_GBS * 1e9) * 1e3 + 2 * (N - 1) * 0.004
                direct = 2 * S / N / (0.65 * XGMI_LINK_GBS * 1e9) * 1e3 + 2 * 0.012
                conv = 0.007 if transport == "bf16" else 0.0                 # two conversion launches (measured)
                row[f"{transport}_N{N}"] = {"t_allreduce_ms": [round(ring + conv, 4), round(direct + conv, 4)],
                                            "efficiency": [round(ms / (ms + t_norm + ring + conv), 3), round(ms / (ms + t_norm + direct + conv), 3)]}
        out[f"configs[{cnum - 1}]"] = row
    return {"model": "efficiency = t_1gpu / (t_1gpu + gradient-norm pass + all-reduce), the all-reduce fully exposed; [pessimistic (ring), "
                     "optimistic (direct reduce-scatter + all-gather over the mesh)], link efficiency 0.65 of 153 GB/s; the driver's N = 2, 4, 8 runs test it",
            "configs": out}


def cpu_baseline(conf=None, seconds=12.0, max_steps=400):
    """The CPU path on this host: oracle train step (same ATen CPU kernels the
    reference dispatches, including the frame classifier it computes and never uses), dropout on, a bounded sample."""
    from oracle import ta3n_oracle as orc
    conf = conf or CONFIGS[2]
    CFG = conf["shape"]
    avg = conf["agg"] == "avgpool"
    cfg = orc.Config(num_class=CFG["C"], num_segments=CFG["T"], feature_dim=CFG["D"], fc_dim=CFG["F"], frame_aggregation=conf["agg"],
                     place_adv=("N", "N", "N") if avg else ("Y", "Y", "Y"), add_loss_DA="none" if avg else "attentive_entropy",
                     use_attn="none" if avg else "TransAttn", compute_dead_branches=True)
    params = synth_state(orc.param_shapes(cfg), seed=7, scale="init")
    xs, xt, ys, yt = synth_batch(CFG["C"], CFG["T"], CFG["D"], CFG["Bs"], CFG["Bt"], seed=1234)
    state = orc.TrainState(params=params, lr=3e-2)
    vdim = cfg.feat_dim if avg else 256
    beta, gamma = ([0.0, 0.0, 0.0], 0.0) if avg else ([0.75, 0.75, 0.5], 0.003)

    def one():
        keep_i, keep_v = 1 - cfg.dropout_i, 1 - cfg.dropout_v
        di = [torch.bernoulli(torch.full((n * CFG["T"], cfg.feat_dim), keep_i)) / keep_i for n in (CFG["Bs"], CFG["Bt"])]
        dv = [torch.bernoulli(torch.full((n, vdim), keep_v)) / keep_v for n in (CFG["Bs"], CFG["Bt"])]
        for _ in range(conf["streams"]):
            orc.train_step(state, xs, xt, ys, beta, gamma, cfg, drop_i=di, drop_v=dv)

    # thread count: the CPU path is many small ATen ops; all hardware threads of a many-core host (or of a cgroup quota) is far slower
    # than a moderate count and the best count differs from box to box (16 on some, 32 on others, a factor 2
    # apart), so probe {16, 32, 64} (what the host has of them; 8 on a small one) - two steps each - time the best, report all.
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    counts = [c for c in (16, 32, 64) if c <= avail] or [min(avail, 8)]
    # every candidate count is TIMED (an equal share of the budget each, at least three steps), the best one is the value and all of them
    # travel in the line: a two-step probe picked the wrong count on a busy host more than once (2.9 k against 5.2 k videos/s on the same box)
    probe, runs = {}, {}
    one()
    for t in counts:
        torch.set_num_threads(t)
        one()
        t1 = time.perf_counter()
        k = 0
        while k < 3 or (k < max_steps and time.perf_counter() - t1 < seconds / len(counts)):
            one()
            k += 1
        runs[t] = (k, time.perf_counter() - t1)
        probe[t] = 1e3 * runs[t][1] / k
    cores = min(probe, key=probe.get)
    torch.set_num_threads(cores)
    n, dt = runs[cores]
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    shape = f"{CFG['Bs']}+{CFG['Bt']} videos, fp32, dropout 0.5" + (", both streams" if conf["streams"] > 1 else "")
    host = f"{avail} hw threads visible of '{model}'"
    out = dict(value=(CFG["Bs"] + CFG["Bt"]) * n / dt, unit="videos/s", cores=cores, kind="port", ms_per_step=1e3 * dt / n,
               probe_ms_per_step_by_threads={str(k): round(v, 2) for k, v in probe.items()},
               sample=f"{n} full train steps ({shape}) of oracle/ta3n_oracle.py (the CPU restatement of the reference's main.train + VideoModel: the "
                      f"same ATen CPU ops, incl. the frame classifier the reference computes and never uses) on {cores} torch threads (the best of "
                      f"{sorted(probe)}, each timed for {seconds / len(counts):.0f} s; {host}) = {1e3 * dt / n:.1f} ms/step")
    # OPT-IN second figure: the reference's own main.train, when somebody staged it explicitly (oracle/reference_runner.py --stage: files
    # verified against pinned sha256 values; never done by build()).  `value` above keeps ONE definition on every box.
    cnum = next(k for k, v in CONFIGS.items() if v is conf)
    try:
        from oracle import reference_runner as rr
        if rr.available():
            import subprocess
            env = dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
            r = subprocess.run([sys.executable, "-m", "oracle.reference_runner", "--config", str(cnum), "--threads", ",".join(map(str, counts)),
                                "--seconds", str(seconds)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
            line = next((ln for ln in r.stdout.splitlines() if ln.startswith("REFERENCE_JSON ")), None)
            if line is None:
                raise RuntimeError(("rc %d: " % r.returncode) + (r.stderr or r.stdout)[-300:])
            ref = json.loads(line[len("REFERENCE_JSON "):])
            out["reference"] = dict(value=ref["videos_per_s"], unit="videos/s", cores=ref["threads"], ms_per_step=ref["ms_per_step"], steps=ref["steps"],
                                    probe_ms_per_step_by_threads=ref["probe_ms_per_step_by_threads"], files_sha256=ref["sha256"], torch=ref["torch"],
                                    what="the reference's own main.train + models.VideoModel from an explicitly staged, sha256-pinned copy (oracle/_ref/py), "
                                         "same synthetic tensors, in a process of its own")
            out["reference_over_port"] = ref["ms_per_step"] / out["ms_per_step"]
        else:
            out["reference"] = None      # nothing staged (the default): profiles/reference_vs_port_cpu.json holds the ratio measured where the checkout is
    except Exception as ex:      # noqa: BLE001 - the port's number stands
        out["reference"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
    return out


def probe_exchanges(build_engines, beta, gamma, lr, fence, world, dev, rank, steps=12, warm=4):
    """Times the pipelined train step under each gradient exchange ("allreduce": one RCCL ncclAllReduce after the last launch;
    "allreduce_overlapped": everything but the shared frame FC's gradient reduced on a second stream WHILE the last launch computes that
    gradient, the rest after it; "sharded": RCCL reduce-scatter, own-shard clip + SGD, all-gather; "peer": the in-tree two-shot all-reduce
    over peer-mapped buffers) and without any (every rank skips
    it: the numbers trained on are then wrong, the time is what is wanted): `steps` steps after `warm`, fenced, MAX over ranks.
    Returns ({candidate: {...}}, fastest candidate).  Every rank takes the same decisions: availability is what the engine
    constructors agreed on collectively, times are all-reduced, ties go to the earlier (simpler) candidate."""
    table, best, best_ms = {}, "allreduce", None

    def timed(eng, n):
        eng.train_steps([(beta, gamma, lr)] * warm)      # (one library call where the engine has one for its exchange, else step by step)
        eng.flush()
        fence()
        t0 = time.perf_counter()
        eng.train_steps([(beta, gamma, lr)] * n)
        eng.flush()
        fence()
        t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        if world > 1:
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        return 1e3 * t.item() / n

    # Which candidates: all four when bench.py started the ranks itself (a job that dies in the probe is started again on the plain
    # all-reduce: launch_ranks) or when asked (TA3N_BENCH_PROBE=all); under a CALLER's launcher only the two RCCL all-reduce schedules -
    # what cannot be retried must not be risked on transports that have only ever run in 1-rank groups and on shared devices.
    want = os.environ.get("TA3N_BENCH_PROBE", "all" if os.environ.get("TA3N_BENCH_OWN_LAUNCHER") == "1" or world == 1 else "safe")
    cands = ("allreduce", "allreduce_overlapped", "sharded", "peer") if want == "all" else ("allreduce", "allreduce_overlapped")
    for cand in ("sharded", "peer"):
        if cand not in cands:
            table[cand] = {"not_probed": "under a caller's launcher only the RCCL all-reduce schedules are probed (TA3N_BENCH_PROBE=all, or the bare "
                                         "`python bench.py --gpus N`, probes every exchange)"}
    for cand in cands:
        row = {}
        try:
            eng = build_engines(cand)[0]
            got = "sharded" if eng._sharded else "peer" if eng.peer is not None else "allreduce_overlapped" if eng._ddp_buckets == 2 else "allreduce"
            if got != cand:
                row["unavailable"] = "the engine kept the default exchange" + (f" ({eng.comm_fallback})" if eng.comm_fallback else "")
            else:
                row["ms_per_step"] = timed(eng, steps)
                row["bytes"] = eng.plan.live_floats * (2 if eng._g16 is not None else 4)
                row["through"] = ("C-ABI RCCL communicator" if eng.comm is not None else "torch.distributed") if cand != "peer" else "csrc/ta3n_peer.hip"
                # sound on EVERY rank, or rejected on every rank: a rank whose peer wait gave up (sticky error word, NaN from then on) must
                # not leave the others inside a collective it skipped - the verdict is local, the decision an all-reduce
                sound = bool(torch.isfinite(eng.P).all().item())
                why = "non-finite parameters after the probe"
                if cand == "peer":
                    try:
                        eng.check_exchange()
                    except Exception as ex:      # noqa: BLE001
                        sound, why = False, f"{ex}"[:160]
                fin = torch.tensor([float(sound)], device=dev)
                if world > 1:
                    torch.distributed.all_reduce(fin, op=torch.distributed.ReduceOp.MIN)
                if fin.item() != 1.0:
                    row["rejected"] = why if not sound else "another rank's exchange failed or delivered non-finite parameters"
                elif best_ms is None or row["ms_per_step"] < best_ms:
                    best, best_ms = cand, row["ms_per_step"]
                if cand == "allreduce":      # the same engine once more without its collective: what each exchange leaves exposed
                    eng.skip_collective = True
                    table["none"] = {"ms_per_step": timed(eng, steps), "what": "the same step with the exchange skipped on every rank"}
                    eng.skip_collective = False
            del eng
        except Exception as ex:      # noqa: BLE001 - a candidate that cannot run is a row of the table, not the end of the run
            row["error"] = f"{type(ex).__name__}: {ex}"[:200]
        table[cand] = row
        gc.collect()
        torch.cuda.empty_cache()
    base = table.get("none", {}).get("ms_per_step")
    for cand, row in table.items():
        if base is not None and "ms_per_step" in row and cand != "none":
            row["exposed_us_per_step"] = round(1e3 * (row["ms_per_step"] - base), 2)
    return {"steps": steps, "warmup": warm, "candidates": table, "chosen": best,
            "what": "the pipelined train step under each gradient exchange, MAX over ranks; the timed region runs on `chosen`"}, best


# tests/test_gpu_bench_two_ranks.py: the WHOLE N > 1 path of this file - launcher, rank checks, exchange probe, MAX-over-ranks timing, the
# line - with two ranks that SHARE cuda:0 over gloo (RCCL does not put two ranks on one device; the one-GPU boxes this is built on have no
# other way to run world_size 2 on hardware).  The line says so (`shared_gpu_test`), its numbers are not a measurement of two GPUs.
SHARED_GPU_TEST = os.environ.get("TA3N_BENCH_SHARED_GPU") == "1"
LAUNCH_TEST = os.environ.get("TA3N_BENCH_LAUNCH_TEST") == "1"      # tests/test_bench_launcher.py: the launcher and the rank handshake on CPU (gloo), no engine, no measurement


def launch_ranks(n_gpus: int) -> int:
    """`python bench.py --gpus N` with N > 1 and no RANK in the environment: this process is not a rank, it is the launcher.  One
    rank per GPU under torch.distributed.run (the command the docstring names), same argv, rendezvous on 127.0.0.1 and a free
    port; returns the job's exit code.  The reference needs no launcher either (main.py:79: one process, nn.DataParallel over
    every visible GPU), so its replacement's benchmark must not need one.  Fails before spawning anything when the box has
    fewer than N GPUs."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n_gpus and not LAUNCH_TEST and not (SHARED_GPU_TEST and have >= 1):
        print(f"[bench] --gpus {n_gpus}: this box has {have} visible GPU(s); refusing to run a {n_gpus}-rank job on fewer devices "
              f"(one process per GPU, RCCL does not share a device between ranks)", file=sys.stderr, flush=True)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL / peer-mapped buffers across processes (must precede HIP initialisation in every rank)
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    print("[bench] launching " + " ".join(cmd[1:8]) + " bench.py " + " ".join(sys.argv[1:]), file=sys.stderr, flush=True)
    env["TA3N_BENCH_OWN_LAUNCHER"] = "1"      # the ranks know a failed job can be started again: they probe EVERY exchange (probe_exchanges)
    rc = subprocess.call(cmd, env=env)
    if rc != 0 and "--exchange" not in sys.argv and not LAUNCH_TEST:
        # The automatic probe includes exchanges that only ever ran on one-GPU boxes (the peer-mapped transport: 1-rank groups and two
        # processes sharing a device).  If the job died, the measurement is still owed: once more on the plain RCCL all-reduce.
        print(f"[bench] the {n_gpus}-rank job exited with {rc}; once more with --exchange allreduce (one RCCL ncclAllReduce per step, no probe)",
              file=sys.stderr, flush=True)
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            cmd[cmd.index("--master-port") + 1] = str(s.getsockname()[1])
        rc = subprocess.call(cmd + ["--exchange", "allreduce"], env=env)
    return rc


def rank_environment(n_gpus: int):
    """(world, rank, local_rank) of this process, checked against --gpus: the line's n_gpus is the number of ranks that really
    came up, so a mismatch between the flag and the job is an error, never a silently smaller run."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != n_gpus:
        raise SystemExit(f"[bench] --gpus {n_gpus} but this job has WORLD_SIZE={world}: launch `python bench.py --gpus {n_gpus}` bare (it "
                         f"starts its own ranks) or give torch.distributed.run --nproc-per-node {n_gpus}")
    if not LAUNCH_TEST and torch.cuda.device_count() < (1 if SHARED_GPU_TEST else world if world > 1 else 1):
        raise SystemExit(f"[bench] rank {rank}: {torch.cuda.device_count()} visible GPU(s) for a {world}-rank job (one process per GPU)")
    return world, rank, local_rank


def launch_test_line(world: int, rank: int) -> None:
    """TA3N_BENCH_LAUNCH_TEST=1 (CPU, gloo): every rank joins the group, a sum all-reduce of ones counts them, rank 0 prints what a
    real line would say about the job's size.  No engine, no timing - `launch_test: true` and `value: null` say so."""
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group("gloo")
        t = torch.ones(1)
        torch.distributed.all_reduce(t)
        seen = int(t.item())
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    else:
        seen = 1
    if seen != world:
        raise SystemExit(f"[bench] {seen} ranks answered in a job of {world}")
    if rank == 0:
        print(json.dumps({"launch_test": True, "value": None, "n_gpus": world, "ranks_seen": seen,
                          "config": {"parallelism": f"dp{world}", "rccl_ranks": None}}), flush=True)


_SAVED_STDOUT = None


def _stdout_to_stderr():
    """From here on everything written to file descriptor 1 - by Python or by any library's printf - goes to stderr."""
    global _SAVED_STDOUT
    if _SAVED_STDOUT is None:
        sys.stdout.flush()
        _SAVED_STDOUT = os.dup(1)
        os.dup2(2, 1)


def _restore_stdout():
    global _SAVED_STDOUT
    if _SAVED_STDOUT is not None:
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)      # the C library's buffered streams (RCCL's banner) - out through stderr before stdout comes back
        except Exception:      # noqa: BLE001
            pass
        os.dup2(_SAVED_STDOUT, 1)
        os.close(_SAVED_STDOUT)
        _SAVED_STDOUT = None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="BASELINE configuration (SURVEY.md 8d numbering); 2 = headline")
    ap.add_argument("--graph", action="store_true", help="replay a captured hipGraph instead of eager launches (eager is "
                    "faster here: 8 launches of 10-70 us each keep the host ahead, a replay adds ~5 us of GPU idle time per step)")
    ap.add_argument("--no-graph", action="store_true", help="(default) eager launches")
    ap.add_argument("--tile", type=int, default=0, help="force a GEMM tile config (114/118/212/122/214/124/221/222)")
    ap.add_argument("--phase-tiles", type=str, default="", help="comma list of per-GEMM-phase tile configs")
    ap.add_argument("--autotune", action="store_true", help="measure tile configs per GEMM launch and use the best")
    ap.add_argument("--xcd", type=int, default=0, help="0/1 XCD-aware tile ordering on, 2 off")
    ap.add_argument("--overlap", action="store_true", help="deferred optimiser update overlapped with the next step's first launch on a "
                    "second stream (measured slower on MI355X: 226 vs 213 us/step - the two cross-stream events cost more than "
                    "the ~9 us of HBM streaming they hide)")
    ap.add_argument("--unfused", action="store_true", help="forward / loss / backward as three calls (15 launches) instead of ta3n_train_step")
    ap.add_argument("--static-hyper", action="store_true", help="diagnostic: do not upload new per-step scalars between replays")
    ap.add_argument("--no-pipeline", action="store_true", help="update at the end of each step + separate scalar upload (ta3n_set_hyper)")
    ap.add_argument("--skip-cpu-baseline", action="store_true")
    ap.add_argument("--no-twins", action="store_true", help="bf16: round fp32 operands in registers everywhere instead of reading bf16 "
                    "twins")
    ap.add_argument("--single-dtype", action="store_true", help="do not also time the other arithmetic (N = 1 runs both by default)")
    ap.add_argument("--dtype", choices=("f32", "bf16", "f32x3"), default=None,
                    help="arithmetic of the contractions: f32 = fp32 MFMA (BASELINE configs[2]); bf16 = operands rounded to bf16, "
                         "bf16 MFMA, fp32 accumulation and fp32 parameters / optimiser state (configs[1]); default: the configuration's")
    ap.add_argument("--wgrads-late", type=int, default=0, help="A/B: TRN weight gradients in the last launch instead of the launch of the F1 gradient (measured slower)")
    ap.add_argument("--serial-streams", action="store_true", help="two-stream configuration: step the two models one after the other on one "
                    "HIP stream instead of concurrently on two")
    ap.add_argument("--plan-heuristic", action="store_true", help="A/B: the plan builder's own tile choice instead of ta3n_amd/tuning.py")
    ap.add_argument("--per-step-calls", action="store_true", help="A/B: one host call per step instead of one ta3n_train_steps call for "
                    "the whole timed region")
    ap.add_argument("--grad-transport", choices=("fp32", "bf16"), default="fp32", help="N > 1: what the gradient all-reduce moves - fp32 (default: "
                    "the exact sum up to the reduction order), or bf16 (every rank's gradients rounded to bf16 and summed in bf16: half the xGMI bytes; "
                    "stated in the line)")
    ap.add_argument("--exchange", choices=("auto", "allreduce", "allreduce_overlapped", "sharded", "peer"), default="auto", help="N > 1: the gradient exchange of the timed "
                    "region - auto (default): the fastest of the three as measured during warm-up (config.exchange_probe names all times); "
                    "allreduce: one RCCL ncclAllReduce per step; allreduce_overlapped: two, the first beside the last launch; sharded: RCCL reduce-scatter + own-shard update + all-gather; peer: the in-tree "
                    "two-shot all-reduce over peer-mapped buffers")
    ap.add_argument("--no-other-configs", action="store_true", help="do not add the 20-step timings of configs[0] / [3] / [4] to the line")
    ap.add_argument("--no-fresh-batch", action="store_true", help="skip the second timed loop with a new device-gathered batch per step")
    ap.add_argument("--phase-reps", type=int, default=20)
    args = ap.parse_args()
    conf = CONFIGS[args.config]
    SH = conf["shape"]
    if args.dtype is None:
        args.dtype = conf["dtype"]
    headline = args.config in (2, 3)
    n_streams = conf["streams"]

    if args.gpus < 1:
        raise SystemExit("[bench] --gpus must be >= 1")
    if args.gpus > 1 and "RANK" not in os.environ:      # bare `python bench.py --gpus N`: start the N ranks ourselves
        raise SystemExit(launch_ranks(args.gpus))
    world, rank, local_rank = rank_environment(args.gpus)
    if LAUNCH_TEST:
        launch_test_line(world, rank)
        return
    if SHARED_GPU_TEST:
        local_rank = 0
    _stdout_to_stderr()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    selftest = os.environ.get("TA3N_DDP_SELFTEST") == "1" and "RANK" in os.environ   # N > 1 code path on 1 rank
    if world > 1 or selftest:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if SHARED_GPU_TEST:
            torch.distributed.init_process_group("gloo")
        else:
            torch.distributed.init_process_group("nccl", device_id=dev)      # "nccl" is RCCL on ROCm

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1 or selftest:
            torch.distributed.barrier()
            torch.cuda.synchronize(dev)

    def run(dtype, steps, warmup, conf=conf, brief=False):
        """Build the engine(s) for one arithmetic, time `steps` train steps after `warmup`; returns the numbers of the JSON line.
        brief: only the step time and the whole-step bound (the other BASELINE configurations inside the default line)."""
        SH = conf["shape"]
        n_streams = conf["streams"]
        headline = conf is CONFIGS[2] or conf is CONFIGS[3]
        bf16 = dtype == "bf16"
        split = dtype == "f32x3"       # fp32-grade contractions as three bf16 MFMAs on operands split hi + lo in registers (TA3N_FLAG_F32_SPLIT)
        twins = (bf16 or split) and not args.no_twins      # (f32x3: "pair twins" - hi and lo planes stored by the producers)
        # None: the measured per-launch choices of ta3n_amd/tuning.py for this shape (what TrainEngine uses by default)
        phase_tiles = [int(v) for v in args.phase_tiles.split(",") if v] or None
        if args.autotune:
            from ta3n_amd.engine import ALL_FLAGS, autotune_phase_tiles
            from ta3n_amd import _lib
            phase_tiles, _ = autotune_phase_tiles(SH["Bs"], SH["Bt"], SH["T"], SH["D"], SH["F"], SH["C"], device=dev,
                                                  flags=ALL_FLAGS | (_lib.FLAG_BF16_MFMA if bf16 else 0) |
                                                  (_lib.FLAG_BF16_STORE if twins else 0) | (_lib.FLAG_F32_SPLIT if split else 0),
                                                  candidates=(114, 118, 212, 122, 214, 124, 221, 222), verbose=(rank == 0))
        if args.tile:
            phase_tiles = []
        if args.plan_heuristic:
            phase_tiles = [0]
        avg = conf["agg"] == "avgpool"
        lr0, gamma, beta = 3e-2, (0.0 if avg else 0.003), ([0.0, 0.0, 0.0] if avg else [0.75, 0.75, 0.5])
        total_steps = 30 * 12                                              # 30 epochs x ~11 steps (1438/128), main.py:334-335

        def build_engines(exchange):
            """The engine(s) of this arithmetic with one gradient exchange (N > 1): None = whatever the environment selects (default: ONE
            RCCL ncclAllReduce per step), "allreduce" / "sharded" (RCCL reduce-scatter, own-shard clip + SGD, all-gather) / "peer" (the
            in-tree two-shot all-reduce over peer-mapped buffers)."""
            kw = {} if exchange is None else dict(sharded_update=exchange == "sharded", peer_exchange=exchange == "peer",
                                                  ddp_buckets=2 if exchange == "allreduce_overlapped" else 1)
            if n_streams == 1 and (world > 1 or selftest):
                kw["share_comm"] = True      # the probe's engines and the timed one share ONE RCCL communicator (one ncclCommInitRank per process)
            es = [TrainEngine(SH["Bs"], SH["Bt"], SH["T"], SH["D"], SH["F"], SH["C"], dropout_i=0.5, dropout_v=0.5,
                              clip=20.0, device=dev, tile_config=args.tile, phase_tiles=phase_tiles, xcd_aware=args.xcd,
                              fused=not args.unfused, bf16=bf16, bf16_store=twins, wgrads_late=args.wgrads_late, aggregation=conf["agg"],
                              f32_split=split, grad_transport=args.grad_transport, use_bn=conf.get("use_bn", "none"), **kw)
                  for _ in range(n_streams)]
            for k, e in enumerate(es):
                shapes = {n: s for n, _, s, _ in e.plan.params}
                e.load_state(synth_state(shapes, seed=7 + k, scale="init"))       # reference init: N(0, 0.001), zero bias
                xs, xt, ys, yt = synth_batch(SH["C"], SH["T"], SH["D"], SH["Bs"], SH["Bt"], seed=1234 + rank + 100 * k)
                e.set_batch(xs.to(dev), xt.to(dev), ys.to(dev))
                e.set_hyper(beta, gamma, lr0)
            return es

        # N > 1: which exchange?  Measured, not assumed: during warm-up every candidate runs the real step on the
        # real message (the flat live-gradient prefix of this configuration) for a few steps, MAX over ranks; the timed region then
        # runs on the fastest, and the line names all of them (config.exchange_probe).  A candidate that cannot be set up on every
        # rank (the constructors agree on that collectively) or that delivers a non-finite parameter is left out; the default
        # exchange always stands as the fallback.  --exchange pins one.
        exchange_probe, chosen_exchange = None, None
        if (world > 1 or selftest) and not brief and n_streams == 1 and not args.unfused and not args.graph:
            if args.exchange == "auto":
                exchange_probe, chosen_exchange = probe_exchanges(build_engines, beta, gamma, lr0, fence, world, dev, rank)
            else:
                chosen_exchange = args.exchange
        engs = build_engines(chosen_exchange)
        eng = engs[0]
        if args.graph:
            for e in engs:
                e.capture()
        deferred = eng.fused and not args.graph and args.overlap
        # default: the update of step n is the first launch of step n+1 and carries that step's scalars (no per-step
        # host-to-device copy); the timed region ends with flush(), so it contains exactly `steps` updates
        pipelined = eng.fused and not args.graph and not deferred and not args.no_pipeline

        # two-stream: the two models are independent - each steps on a HIP stream of its own, so one model's launches fill the CUs
        # and the launch / prologue latencies the other leaves idle (--serial-streams: one after the other on one stream)
        side = [torch.cuda.Stream(dev) for _ in engs] if (n_streams > 1 and not args.serial_streams) else None
        if side:
            for s_ in side:
                s_.wait_stream(torch.cuda.current_stream(dev))

        def flush_all():
            for k, e in enumerate(engs):
                if side:
                    with torch.cuda.stream(side[k]):
                        e.flush()
                else:
                    e.flush()

        def step(i):
            p = float(i % total_steps) / total_steps
            lr = lr0 if i == 0 else lr_dann(lr0, p)
            if side:
                for e, s_ in zip(engs, side):
                    with torch.cuda.stream(s_):
                        (e.train_step_pipelined if pipelined else e.train_step)(beta, gamma, lr)
                return
            for e in engs:                                                 # (or: both models step, one after the other)
                if args.static_hyper and e.graph is not None:
                    e.graph.replay()
                elif deferred:
                    e.train_step_deferred(beta, gamma, lr)
                elif pipelined:
                    e.train_step_pipelined(beta, gamma, lr)
                else:
                    e.train_step(beta, gamma, lr)

        # Default at N = 1: the K steps are enqueued by ONE call into the library (ta3n_train_steps; the schedule of beta / lr /
        # dropout seeds is evaluated ahead of time and travels by value) - the host is then off the step's critical path, which
        # under the 20-step protocol on a slow host core was 21 % of the step.  --per-step-calls: one call per step.
        batched = (pipelined and (not side and len(engs) == 1 or n_streams > 1) and not args.per_step_calls and
                   ((world == 1 and not selftest) or (eng.comm is not None and eng._ddp_buckets == 1)))
        two = None
        if batched and n_streams > 1:       # configs[4]: both models' steps from ONE ta3n_train_steps_multi call (two HIP streams, or one)
            from ta3n_amd.two_stream import TwoStreamEngine
            two = TwoStreamEngine.__new__(TwoStreamEngine)
            two.streams, two._hip_streams = engs, side

        def sched(i0, n):
            out = []
            for i in range(i0, i0 + n):
                p = float(i % total_steps) / total_steps
                out.append((beta, gamma, lr0 if i == 0 else lr_dann(lr0, p)))
            return out

        def run_steps(i0, n):
            if two is not None:
                two.train_steps(sched(i0, n))
            elif batched:
                eng.train_steps(sched(i0, n))
            else:
                for i in range(i0, i0 + n):
                    step(i)

        gc.collect()                                 # before the warmup, not between warmup and timing: a collection takes tens of
        gc.disable()                                 # milliseconds during which the GPU would idle and drop its clocks; disabled while
        run_steps(0, warmup)                         # timing - eager launches: a collector pause on the host would show up as GPU idle time
        flush_all()
        fence()
        t0 = time.perf_counter()
        run_steps(warmup, steps)
        flush_all()                                  # the K-th update is inside the timed region
        fence()
        elapsed = time.perf_counter() - t0
        gc.enable()
        if world > 1:
            t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            elapsed = t.item()
        # a two-stream step processes each video through both models: videos/s counts videos, not model passes
        res = {"ms_per_step": 1e3 * elapsed / steps, "value": (SH["Bs"] + SH["Bt"]) * world * steps / elapsed}
        if batched and two is None and not brief and world == 1 and not selftest and eng.can_batch_steps() and not args.no_fresh_batch:
            try:
                # The same K steps once more with a DIFFERENT batch every step: a synthetic dataset resident in HBM
                # as a packed feature store (ta3n_amd/feature_store.py: 640 videos x 16-47 frames), random video ids per step, each step's
                # batch assembled on the device by the gather kernel that the same ta3n_train_steps call enqueues in front of it (ta3n_feed:
                # test-mode segment indices of dataset.py:103-116 + row copy into the input buffer and its bf16 twin).  `value` keeps the
                # resident-batch protocol of SURVEY 8(d); this is what a training loop pays.
                from ta3n_amd.feature_store import FeatureStore
                g = torch.Generator(device="cpu").manual_seed(4321 + rank)
                nf = torch.randint(16, 48, (640,), generator=g)
                rows = torch.randn(int(nf.sum()), SH["D"], device=dev).abs_()
                store_bf16 = bool(bf16 and twins)      # the bf16 arithmetic reads the input's bf16 twin only: a store packed in bf16 (feature_store.pack(dtype="bf16"))
                if store_bf16:                         # feeds it directly - the same operands (RNE of the fp32 features), half the gather bytes
                    rows = rows.to(torch.bfloat16).view(torch.int16)
                store = FeatureStore.from_tensors(rows, nf.to(dev), torch.randint(0, SH["C"], (640,), generator=g).to(dev))
                ids = [torch.randint(0, 640, (warmup + steps, n_), generator=g, dtype=torch.int32).to(dev) for n_ in (SH["Bs"], SH["Bt"])]
                eng.train_steps(sched(warmup + steps, warmup), feeds=((store, ids[0][:warmup]), (store, ids[1][:warmup])))
                flush_all()
                fence()
                t1 = time.perf_counter()
                eng.train_steps(sched(2 * warmup + steps, steps), feeds=((store, ids[0][warmup:]), (store, ids[1][warmup:])))
                flush_all()
                fence()
                e_f = time.perf_counter() - t1
                if world > 1:
                    t = torch.tensor([e_f], device=dev, dtype=torch.float64)
                    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
                    e_f = t.item()
                res["fresh_batch"] = {"ms_per_step": 1e3 * e_f / steps, "value": (SH["Bs"] + SH["Bt"]) * world * steps / e_f,
                                      "what": "the timed loop repeated with a new batch per step, gathered on the device from a packed feature store "
                                              "resident in HBM (640 synthetic videos, 16-47 frames each, packed as " + ("bf16: the gather writes the input's bf16 twin" if store_bf16 else "fp32") + "; ta3n_feed inside the same ta3n_train_steps call)"}
                del store, rows, ids
            except Exception as ex:      # noqa: BLE001 - an extra figure must not cost the line (nor, at N > 1, the scaling run)
                res["fresh_batch_error"] = f"{type(ex).__name__}: {ex}"[:200]
        if (world > 1 or selftest) and not brief:
            # what the gradient exchange costs per step: the same loop once more WITHOUT the collective (every rank skips it; the
            # numbers it trains on are then wrong, the timing is what is wanted) - the difference is the exposed collective time
            eng.skip_collective = True
            run_steps(warmup + steps, min(warmup, 5))
            flush_all()
            fence()
            t1 = time.perf_counter()
            run_steps(warmup + steps + 5, steps)
            flush_all()
            fence()
            e2 = time.perf_counter() - t1
            eng.skip_collective = False
            if world > 1:
                t = torch.tensor([e2], device=dev, dtype=torch.float64)
                torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
                e2 = t.item()
            res["collective"] = {"exposed_us_per_step": 1e6 * (elapsed - e2) / steps, "step_without_collective_ms": 1e3 * e2 / steps,
                                 "bytes": eng.plan.live_floats * (2 if eng._g16 is not None else 4),
                                 "what": "the timed loop repeated with the all-reduce skipped on every rank; exposed = difference per step"}
        if rank != 0:
            return res
        wsb = whole_step_bound(conf, dtype, eng)
        res["whole_step"] = {**wsb, "frac": wsb["bound_us"] / (1e3 * res["ms_per_step"]),
                             "what": "SURVEY.md 8(d): max(algorithmic FLOPs / MFMA peak, algorithmic bytes / 6.3 TB/s) per step, divided by the measured step time"}
        if brief:
            if eng.fused and conf["agg"] == "trn-m":      # per-launch figures of the tile-list GEMM for the other configurations too
                res["gemm_launches"] = gemm_launch_table(eng, eng.time_phases(max(3, args.phase_reps // 2)), bf16, split)
            return res
        res["finite"] = all(bool(torch.isfinite(e.P).all().item()) for e in engs)
        if eng._sharded:
            res["gradient_exchange"] = ("sharded update: RCCL reduce-scatter (region B beside the last launch), own-shard clip + SGD, all-gather of the parameters "
                                        "(region B beside the next step's first launch); " + ("bf16" if eng._g16 is not None else "fp32") + " gradient transport"
                                        if eng.comm is not None else "sharded update over torch.distributed")
        elif eng.peer is not None:
            res["gradient_exchange"] = ("two-shot all-reduce over peer-mapped buffers (csrc/ta3n_peer.hip), " +
                                        ("bf16" if eng.peer.bf16 else "fp32") + " transport")
        elif eng.comm is not None:
            res["gradient_exchange"] = ("RCCL ncclAllReduce from the C ABI on the step's stream, " +
                                        ("bf16" if eng._g16 is not None else "fp32") + " transport")
        else:
            res["gradient_exchange"] = ((f"torch.distributed all_reduce (backend {torch.distributed.get_backend()}), fp32" if SHARED_GPU_TEST else
                                         "torch.distributed all_reduce (backend nccl = RCCL), fp32") +
                                        (f" [C-ABI communicator unavailable: {eng.comm_fallback}]" if eng.comm_fallback else ""))
        # the number of ranks the exchange really spans: the library's communicator, else the torch.distributed (RCCL) group
        res["rccl_ranks"] = (int(eng._L.ta3n_comm_world(eng.comm.handle)) if eng.comm is not None else
                             (torch.distributed.get_world_size() if (world > 1 or selftest) else None))
        if (world > 1 or selftest) and res["rccl_ranks"] != world:
            raise SystemExit(f"[bench] the gradient exchange spans {res['rccl_ranks']} rank(s) in a job of {world}")
        res["exchange_probe"] = exchange_probe
        res["exchange"] = chosen_exchange
        res["deferred"] = deferred
        res["batched"] = batched
        res["pipelined"] = pipelined
        res["fused"] = eng.fused
        # live per-launch timing of the dominant kernel (the tile-list GEMM), HIP events on the launch stream
        phases = eng.time_phases(args.phase_reps)
        gemm_plain_ms = sum(p[3] for p in phases if p[0] == 0)       # the six GEMM launches without the update riding in the first
        side_update = False
        if pipelined and eng._side_update and world == 1 and not selftest:
            # the timed loop ran ta3n_train_step_after_update: its first two launches are the shared-FC update and the first GEMM
            # launch WITH the rest of the update as side workgroups - time those (not the plain first launch + a whole-buffer SGD)
            upd_ms, f1_ms = eng.time_update_launches(args.phase_reps)
            first_gemm = next(i for i, p in enumerate(phases) if p[0] == 0)
            phases = [(5, 0, 0, upd_ms) if p[0] == 5 else p for p in phases]
            phases[first_gemm] = (0, phases[first_gemm][1], phases[first_gemm][2] + 256, f1_ms)
            side_update = True
        gemm = [p for p in phases if p[0] == 0]
        gemm_ms = sum(p[3] for p in gemm)
        flops = algorithmic_flops(conf)
        tflops = flops / (gemm_ms * 1e-3) / 1e12
        traffic, traffic_src = measured_traffic(dtype) if (headline and eng.fused and (twins or not bf16)) else (None, {"file": None})
        extra = {"kernel": f"ta3n::gemm_tiles ({len(gemm)} launches/step" + (" and stream)" if n_streams > 1 else ")"), "launches": len(gemm),
                 "flops_per_launch": flops / max(len(gemm), 1), "avg_launch_us": 1e3 * gemm_ms / max(len(gemm), 1),
                 "all_kernels_us": 1e3 * sum(p[3] for p in phases), "traffic": traffic,
                 "traffic_unit": "bytes per GEMM launch: rocprofv3 PMC FETCH_SIZE*2 + WRITE_SIZE (tools/measure_traffic.py; the PMC passes "
                                 "run the step with the update as its own kernel, so this compares with gemm_only.bytes_per_launch)",
                 "traffic_source": traffic_src,
                 "per_phase_us": [[p[0], p[1], p[2], round(1e3 * p[3], 2)] for p in phases],
                 "per_phase_note": "[kind (0 GEMM, 5 optimiser, 6 heads), tile, workgroups, us]; HIP events on the launch stream" +
                 ("; the optimiser entry is the shared-FC update that opens the step, the first GEMM entry includes the 256 side "
                  "workgroups that apply the rest of the update" if side_update else "")}
        if split:          # three bf16 MFMAs per product block: the matrix-core ceiling for these FLOPs is a third of the bf16 peak
            peak = PEAK_BF16_MFMA_TFLOPS / 3
            res["roofline"] = {"bound": "mfma", "achieved": tflops, "peak": peak, "unit": "TFLOP/s", "frac": tflops / peak,
                               "peak_note": "bf16 MFMA peak / 3 (a_hi b_hi + a_hi b_lo + a_lo b_hi per product)", **extra}
        elif not bf16:     # fp32 MFMA: 95 FLOP/B against a machine balance of 25 -> MFMA-bound
            res["roofline"] = {"bound": "mfma", "achieved": tflops, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                               "frac": tflops / PEAK_FP32_MFMA_TFLOPS, **extra}
        else:              # bf16 MFMA makes the math 16x cheaper than fp32's: the binding roofline is HBM
            gemm_bytes = algorithmic_gemm_bytes_bf16(**SH, agg=conf["agg"])
            # ONE definition across rounds: `frac` = algorithmic bytes of the CONTRACTIONS per launch / the average duration of
            # the kernel's launches without any rider (ta3n_time_phases) - round 1's figure.  In the pipelined step the first launch also
            # carries the optimiser update of every parameter but the shared frame FC (256 side workgroups, 20 B per parameter); that
            # inclusive figure (round 2's `frac`) is published beside it as `with_update`.
            upd_params = sum(math.prod(s_) for n_, _, s_, live in eng.plan.params
                             if live and not n_.startswith("fc_feature_shared_source")) if side_update else 0
            nbytes = gemm_bytes + 20 * upd_params
            gbs = nbytes / (gemm_ms * 1e-3) / 1e9
            gbs_plain = gemm_bytes / (gemm_plain_ms * 1e-3) / 1e9
            extra["avg_launch_us"] = 1e3 * gemm_plain_ms / max(len(gemm), 1)
            res["roofline"] = {"bound": "hbm", "achieved": gbs_plain, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs_plain / HBM_PEAK_GBS,
                               "bytes_per_launch": gemm_bytes / max(len(gemm), 1),
                               "bytes_note": "algorithmic bytes of the contraction launches: input once as bf16, live weights 2 + 2 B read "
                               "(forward + backward) + 4 B gradient written = %.2f MB per step, over %d launches" % (gemm_bytes / 1e6, len(gemm)),
                               "with_update": {"bytes_per_launch": nbytes / max(len(gemm), 1), "avg_launch_us": 1e3 * gemm_ms / max(len(gemm), 1),
                                               "achieved": gbs, "frac": gbs / HBM_PEAK_GBS,
                                               "what": "the same launches as the timed step runs them: the first one also applies the optimiser "
                                                       "update of %d parameters (20 B each = %.2f MB) - round 2's `frac`" % (upd_params, 20 * upd_params / 1e6)},
                               "mfma_tflops": tflops, "mfma_frac_of_bf16_peak": tflops / PEAK_BF16_MFMA_TFLOPS, **extra}
            if wsb["bound"] == "mfma":      # (configs[3]: 59.6 us of bf16 MFMA against 51 us of bytes - the matrix cores are the binding roofline there)
                res["roofline"].update({"bound": "mfma", "achieved": tflops, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                                        "frac": tflops / PEAK_BF16_MFMA_TFLOPS, "hbm_gbs": gbs_plain, "hbm_frac": gbs_plain / HBM_PEAK_GBS})
        res["roofline"]["whole_step"] = res["whole_step"]
        if eng.fused:
            res["roofline"]["gemm_launches"] = gemm_launch_table(eng, [p for p in eng.time_phases(args.phase_reps)], bf16, split)
        # tile code per GEMM launch as the plan built it: WM WN WK + 1000 x (LDS stages, + 16: reads bf16 twins) + 100000 x blocking
        # (1: 2 row blocks per wave, 2: 2 column blocks, 3: 2 x 2 - 128x64 / 64x128 / 128x128 tiles)
        res["phase_tiles"] = [ph["tile"] + 100000 * ((ph.get("rm", 1) > 1) + 2 * (ph.get("rn", 1) > 1))
                              for ph in eng.plan.description["phases"] if ph["kind"] == 0 and ph["group"] != 5]
        return res

    main_res = run(args.dtype, args.steps, args.warmup)
    other, third = None, None
    if headline and not args.single_dtype and not selftest:   # the other arithmetic, same process (all ranks): at N > 1 the fp32 line is BASELINE configs[2]
        other = run("f32" if args.dtype == "bf16" else "bf16", max(50, args.steps // 2), max(10, args.warmup // 2))
        if args.dtype != "f32x3" and world == 1:               # and the fp32-grade split arithmetic (same fp32 parity tests as "f32")
            try:
                third = run("f32x3", max(50, args.steps // 2), max(10, args.warmup // 2))
            except Exception as ex:      # noqa: BLE001 - an extra line must not cost the headline line
                print(f"[bench] split-arithmetic run failed: {type(ex).__name__}: {ex}", file=sys.stderr, flush=True)

    # the other BASELINE configurations, bounded (20 steps after 5: well under 2 s each), so that the default line carries
    # driver-visible timings of configs[0] / [3] / [4] with their SURVEY 8(d) bounds
    configs_line = None
    if headline and world == 1 and not args.single_dtype and not selftest and not args.no_other_configs:
        configs_line = {}
        for cnum in (1, 4, 5):
            cf = CONFIGS[cnum]
            try:
                r = run(cf["dtype"], 20, 5, conf=cf, brief=True)
                ws = r["whole_step"]
                configs_line[f"configs[{cnum - 1}]"] = {"workload": cf["name"], "dtype": cf["dtype"], "ms_per_step": r["ms_per_step"],
                                                        "value": r["value"], "unit": "videos/s", "steps": 20, "warmup": 5,
                                                        "bound": ws["bound"], "bound_us": ws["bound_us"], "frac_of_bound": ws["frac"]}
                if "gemm_launches" in r:
                    configs_line[f"configs[{cnum - 1}]"]["gemm_launches"] = r["gemm_launches"]
                # measured HBM-side traffic of this configuration's GEMM launches, where a PMC measurement is committed (VERDICT r05 item 3:
                # the ratio per configuration, not only for the headline) against the algorithmic bytes of its contractions
                tr, tr_src = measured_traffic(cf["dtype"], key=f"configs[{cnum - 1}]")
                if tr is not None:
                    alg = algorithmic_gemm_bytes_bf16(**cf["shape"], agg=cf["agg"]) / max(len(r.get("gemm_launches", [])), 1)
                    configs_line[f"configs[{cnum - 1}]"].update({"traffic": tr, "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": tr / alg,
                                                                 "traffic_source": tr_src})
            except Exception as ex:      # noqa: BLE001 - an extra entry must not cost the headline line
                configs_line[f"configs[{cnum - 1}]"] = {"error": f"{type(ex).__name__}: {ex}"[:200]}
    # SURVEY 8(f)4 on the fused path (round 6): the headline step with use_bn AdaBN - two BatchNorm launches inside ta3n_train_step, 10 launches
    # instead of 8, the running statistics tracked by the BatchNorm launch itself (K steps from one library call, like the headline)
    variants = None
    if headline and world == 1 and not args.single_dtype and not selftest and not args.no_other_configs:
        variants = {}
        try:
            cf = dict(CONFIGS[2], use_bn="AdaBN", name=CONFIGS[2]["name"] + ", use_bn AdaBN (domain BatchNorm behind the shared frame FC) inside the fused step")
            r = run(cf["dtype"], 20, 5, conf=cf, brief=True)
            variants["headline+AdaBN"] = {"workload": cf["name"], "dtype": cf["dtype"], "ms_per_step": r["ms_per_step"], "value": r["value"], "unit": "videos/s",
                                          "steps": 20, "warmup": 5, "launches_per_step": 10}
        except Exception as ex:      # noqa: BLE001 - an extra entry must not cost the headline line
            variants["headline+AdaBN"] = {"error": f"{type(ex).__name__}: {ex}"[:200]}
        # The paper's other baseline rows - ens_DA MCD, dis_DA DAN / JAN - stay unfused launch lists (they couple the two halves of what the
        # heads kernel fuses); since round 6 their loss assembly comes from the library too (ta3n_mcd_*, ta3n_discrepancy), no framework between the
        # launches.  One train_step call per step (several library calls each), host included.
        SHv, bf = CONFIGS[2]["shape"], CONFIGS[2]["dtype"] == "bf16"
        for key, kw in (("headline+MCD", dict(ens_DA="MCD", mu=0.5)), ("headline+DAN", dict(dis_DA="DAN", alpha=0.5)),
                        ("headline+JAN", dict(dis_DA="JAN", alpha=0.5, place_dis=("Y", "Y", "N")))):
            try:
                e = TrainEngine(SHv["Bs"], SHv["Bt"], SHv["T"], SHv["D"], SHv["F"], SHv["C"], dropout_i=0.5, dropout_v=0.5, clip=20.0, device=dev,
                                bf16=bf, bf16_store=bf, **kw)
                e.load_state(synth_state({n: s_ for n, _, s_, _ in e.plan.params}, seed=7, scale="trained"))
                xs, xt, ys, yt = synth_batch(SHv["C"], SHv["T"], SHv["D"], SHv["Bs"], SHv["Bt"], seed=1234 + rank)
                e.set_batch(xs.to(dev), xt.to(dev), ys.to(dev))
                # (trained-scale weights and lr 1e-3, not the headline's N(0, 0.001) initialisation and 3e-2: on ONE repeated synthetic batch the
                #  discrepancy losses' data-dependent bandwidth - the mean pairwise distance of near-identical features, loss.py:55 - drives the step
                #  to NaN within ~25 steps, in the reference's own algebra too: the torch assembly and ta3n_discrepancy go there digit for digit)
                for _ in range(15):      # (the first steps of a configuration pay one-time costs: code objects, allocator)
                    e.train_step([0.75, 0.75, 0.5], 0.003, 1e-3)
                torch.cuda.synchronize(dev)
                gc.collect()             # (as in run(): a collector pause inside a 20-step region of a host-driven loop is tens of milliseconds)
                gc.disable()
                t0 = time.perf_counter()
                for _ in range(20):
                    e.train_step([0.75, 0.75, 0.5], 0.003, 1e-3)
                torch.cuda.synchronize(dev)
                dt = (time.perf_counter() - t0) / 20
                gc.enable()
                variants[key] = {"workload": CONFIGS[2]["name"] + ", " + ", ".join(f"{k}={v}" for k, v in kw.items()), "dtype": CONFIGS[2]["dtype"],
                                 "ms_per_step": 1e3 * dt, "value": (SHv["Bs"] + SHv["Bt"]) / dt, "unit": "videos/s", "steps": 20, "warmup": 15,
                                 "what": "unfused launch lists, the option's loss assembly from the library; one train_step call per step, host included; trained-scale synthetic weights, lr 1e-3",
                                 "finite": bool(torch.isfinite(e.P).all().item())}
                del e
            except Exception as ex:      # noqa: BLE001 - an extra entry must not cost the headline line
                variants[key] = {"error": f"{type(ex).__name__}: {ex}"[:200]}
    if rank == 0:
        from ta3n_amd import tolerances as tol
        arith = {"bf16": "bf16 MFMA on operands rounded to nearest-even, fp32 accumulation, fp32 parameters / gradients / optimiser state "
                         f"(BASELINE configs[1]); parity gate tests/test_gpu_bf16.py against the bf16-operand oracle: logits <= {tol.BF16_LOGIT_REL_RMS:g} of "
                         f"rms, every gradient tensor rel. L2 <= {tol.BF16_GRAD_REL_L2:g} (median <= {tol.BF16_GRAD_REL_L2_MEDIAN:g}); distance from the fp32 "
                         f"reference (tests/test_gpu_gradients.py): logits <= {tol.BF16_REF_LOGIT_REL_RMS:g} of rms, gradient tensors median rel. L2 <= "
                         f"{tol.BF16_REF_GRAD_REL_L2_MEDIAN:g}",
                 "f32": f"fp32 MFMA throughout (BASELINE configs[2] arithmetic); parity: logits within {tol.LOGIT_ATOL:g} of the reference's CPU path, "
                        f"every gradient tensor rel. L2 <= {tol.F32_GRAD_REL_L2:g} (median over tensors <= {tol.F32_GRAD_REL_L2_MEDIAN:g}) "
                        "(tests/test_gpu_parity.py, tests/test_gpu_gradients.py)",
                 "f32x3": "fp32-grade contractions on the bf16 MFMA: operands split hi + lo = bf16(x) + bf16(x - hi) " +
                          ("in registers, " if args.no_twins else "by the producing kernels (hi / lo planes in HBM, nothing converted in the K loops), ") +
                          "a_hi b_hi + a_hi b_lo + a_lo b_hi accumulated in fp32 (~2^-16 per product; not IEEE fp32 multiplication); fp32 "
                          f"parameters, gradients, optimiser; logits within {tol.LOGIT_ATOL:g} of the reference's CPU path, every gradient "
                          f"tensor rel. L2 <= {tol.F32X3_GRAD_REL_L2:g} (median <= {tol.F32X3_GRAD_REL_L2_MEDIAN:g}) (same tests, [f32x3] / [bf16x3] ids)"}
        out = {
            "metric": "src+tgt videos/sec per train step, UCF->HMDB_full 5-seg TA3N" if headline else
                      "src+tgt videos/sec per train step (BASELINE configs[%d])" % (args.config - 1),
            "value": main_res["value"], "unit": "videos/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": main_res["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic", **({"shared_gpu_test": "the ranks of this run SHARE one GPU over gloo: a test of the N > 1 path, "
                                                           "not a measurement of N GPUs"} if SHARED_GPU_TEST else {}),
            "config": {"workload": conf["name"] + ", dropout 0.5/0.5, clip 20, Nesterov SGD; " + arith[args.dtype],
                       "baseline_config": args.config - 1,
                       "global_batch": (SH["Bs"] + SH["Bt"]) * world, "parallelism": f"dp{world}",
                       "launch": "hipGraph" if args.graph else (("eager, all timed steps of both models enqueued by one ta3n_train_steps_multi call, one HIP stream per model"
                                                                 if n_streams > 1 else "eager, all timed steps enqueued by one ta3n_train_steps call")
                                                                if main_res.get("batched") else "eager, one host call per step"),
                       "finite": main_res["finite"],
                       "step": "fused (ta3n_train_step)" if main_res["fused"] else "forward+loss+backward",
                       "update": "deferred: overlaps the next step's first launch" if main_res["deferred"] else
                       ("opens the next step, carrying its scalars; all but the shared frame FC's part rides in that step's first GEMM "
                        "launch (ta3n_train_step_after_update)" if main_res["pipelined"] else "end of step"),
                       "gradient_exchange": None if (world == 1 and not selftest) else main_res.get("gradient_exchange"),
                       "rccl_ranks": main_res.get("rccl_ranks"),
                       "exchange": main_res.get("exchange"),
                       "exchange_probe": main_res.get("exchange_probe"),
                       "collective": main_res.get("collective"),
                       "phase_tiles": main_res["phase_tiles"]},
            "roofline": main_res["roofline"],
        }
        if other is not None:
            o_dtype = "f32" if args.dtype == "bf16" else "bf16"
            o = {"dtype": o_dtype, "what": arith[o_dtype], "value": other["value"], "unit": "videos/s", "ms_per_step": other["ms_per_step"],
                 "roofline": {k: other["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us",
                                                                "per_phase_us")}}
            out["other_arithmetic"] = o
            # the same numbers inside `roofline`, so that the parity-qualified fp32 figure travels with the headline line
            out["roofline"]["other_arithmetic"] = {"dtype": o_dtype, "value": other["value"], "value_unit": "videos/s", "ms_per_step": other["ms_per_step"],
                                                   "whole_step": other["whole_step"],
                                                   **{k: other["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac")}}
        if third is not None:
            out["roofline"]["split_arithmetic"] = {"dtype": "f32x3", "what": arith["f32x3"], "value": third["value"], "value_unit": "videos/s",
                                                   "ms_per_step": third["ms_per_step"], "whole_step": third["whole_step"],
                                                   **{k: third["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac",
                                                                                        "avg_launch_us", "per_phase_us")}}
        if "fresh_batch_error" in main_res:
            out["fresh_batch_error"] = main_res["fresh_batch_error"]
        if "fresh_batch" in main_res:
            out["value_fresh_batch"] = main_res["fresh_batch"]["value"]
            out["ms_per_step_fresh_batch"] = main_res["fresh_batch"]["ms_per_step"]
            out["fresh_batch_note"] = main_res["fresh_batch"]["what"]
            if other is not None and "fresh_batch" in other:
                out["other_arithmetic"]["value_fresh_batch"] = other["fresh_batch"]["value"]
                out["other_arithmetic"]["ms_per_step_fresh_batch"] = other["fresh_batch"]["ms_per_step"]
        if configs_line:
            out["configs"] = configs_line
        if variants:
            out["variants"] = variants
        try:      # projection for every configuration timed in this run (at N > 1: the benched one, from the step without the collective)
            t1 = {}
            if world == 1 and not selftest:
                t1[args.config] = main_res["ms_per_step"]
                if other is not None and headline:
                    t1[3 if args.dtype == "bf16" else 2] = other["ms_per_step"]
                for k_, v_ in (configs_line or {}).items():
                    if "ms_per_step" in v_:
                        t1[int(k_[8:-1]) + 1] = v_["ms_per_step"]
            elif main_res.get("collective"):
                t1[args.config] = main_res["collective"]["step_without_collective_ms"]
            if t1:
                out["config"]["scaling_projection"] = scaling_projection(t1)
                if world > 1:
                    out["config"]["scaling_projection"]["measured_here"] = {"n_gpus": world, "ms_per_step": main_res["ms_per_step"],
                                                                           "exposed_collective_us": main_res["collective"]["exposed_us_per_step"]}
        except Exception as ex:      # noqa: BLE001 - an extra entry must not cost the line
            out["config"]["scaling_projection"] = {"error": f"{type(ex).__name__}: {ex}"[:200]}
        if not args.skip_cpu_baseline and world == 1:       # the CPU path is timed on rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(conf)
        # ONE JSON line on stdout and nothing else: libraries print there too (RCCL's version banner sits in the C stdio buffer until the
        # process exits and would land BEHIND the line), so file descriptor 1 was pointed at stderr for the whole run (_stdout_to_stderr);
        # flush whatever C and Python still hold, give stdout back, print the line
        _restore_stdout()
        print(json.dumps(out), flush=True)
        _stdout_to_stderr()      # (whatever is printed while the process group shuts down is not part of the answer either)
    if world > 1 or selftest:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
