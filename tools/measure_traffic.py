"""HBM-side traffic of the GEMM launches of the fused step (headline shape), per arithmetic, from rocprofv3 PMC passes.

Run on the GPU box (bench.py cannot collect PMC counters inside its timed loop: they need their own rocprofv3 passes):
    cd /tmp && export TMPDIR=/tmp && python $GRAFT_REPO_ROOT/tools/measure_traffic.py $GRAFT_REPO_ROOT/gpurun_out/prof_r02
Writes <out>/gemm_traffic.json (copy to profiles/gemm_traffic.json: bench.py reports it as roofline.traffic together with
the source hash it was measured on) and <out>/pmc_<dtype>.txt (the per-dispatch counter tables of the last fused step).
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half of the bytes of wide coalesced reads
(MI355X_MICROARCH.md, HBM section), hence the factor 2.  Counters are collected in separate passes with --kernel-trace only."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ta3n_amd.build import source_hash  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "prof_r02")
os.makedirs(out, exist_ok=True)
PASSES = ["FETCH_SIZE", "WRITE_SIZE", "SQ_WAVES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"]
result = {"source_hash": source_hash(), "shape": "128+74 videos, T=5, D=2048 (headline)"}
for dtype, cfgnum in (("bf16", 2), ("f32", 2), ("f32x3", 2), ("bf16", 4)):      # (round 6: + configs[3], bench.py --config 4, in its arithmetic)
    key = dtype if cfgnum == 2 else f"configs[{cfgnum - 1}]"
    per_pass = {}
    lines = []
    for p in PASSES:
        tag = p.split()[0]
        d = f"/tmp/pmc_{key.replace('[', '').replace(']', '')}_{tag}"
        subprocess.run(["rm", "-rf", d])
        cmd = ["rocprofv3", "--pmc", *p.split(), "--kernel-trace", "-d", d, "-o", "out", "--output-format", "csv", "--",
               sys.executable, os.path.join(ROOT, "tools", "prof_step.py"), "0", "0", "fused", dtype, str(cfgnum)]
        subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            lines.append(f"## pass {p}: no counter file")
            continue
        rows = list(csv.DictReader(open(files[0])))
        by = collections.OrderedDict()
        for r in rows:
            by.setdefault((int(r["Dispatch_Id"]), r["Kernel_Name"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
        # the last fused step = the GEMM dispatches after the second-to-last optimiser launch
        disp = list(by.items())
        sgd = [i for i, ((_, name), _) in enumerate(disp) if "sgd" in name]
        last = disp[sgd[-2] + 1: sgd[-1] + 1] if len(sgd) >= 2 else disp
        lines.append(f"## pass: {p}")
        for (did, name), v in last:
            short = name.replace("void ", "").replace("ta3n::", "").replace("(anonymous namespace)::", "")[:44]
            w = v.get("SQ_WAVES", 0.0) or 1.0
            lines.append(f"{did} {short} " + " ".join(f"{k}={val:.0f}" if not k.startswith("SQ_") or k == "SQ_WAVES" else f"{k[3:]}={val / w:.0f}/w" for k, val in sorted(v.items())))
        per_pass[tag] = [(name, v) for (did, name), v in last if "gemm_tiles" in name]
    open(os.path.join(out, f"pmc_{dtype}.txt" if cfgnum == 2 else f"pmc_config{cfgnum}_{dtype}.txt"), "w").write("\n".join(lines) + "\n")
    if "FETCH_SIZE" in per_pass and "WRITE_SIZE" in per_pass and per_pass["FETCH_SIZE"]:
        n = len(per_pass["FETCH_SIZE"])
        fetch = sum(v["FETCH_SIZE"] for _, v in per_pass["FETCH_SIZE"]) * 1024
        write = sum(v["WRITE_SIZE"] for _, v in per_pass["WRITE_SIZE"]) * 1024
        result[key] = {"dtype": dtype, "gemm_launches": n, "fetch_bytes_x2": 2 * fetch, "write_bytes": write,
                         "bytes_per_gemm_launch": (2 * fetch + write) / n, "passes": "FETCH_SIZE, WRITE_SIZE (separate rocprofv3 --pmc runs)"}
json.dump(result, open(os.path.join(out, "gemm_traffic.json"), "w"), indent=1)
print(json.dumps(result))
