"""Per-launch cache behaviour of the fused step at one shape, from rocprofv3 PMC passes (each pass its own run, --kernel-trace only):
L2 hits / misses (TCC_HIT_sum, TCC_MISS_sum), fabric-side read bytes (FETCH_SIZE x 2, MI355X_MICROARCH.md), the vector L1's requests to
the L2, LDS bank conflicts, MFMA busy cycles - next to each launch's duration.  The question it answers: is a GEMM launch bound by the
fabric (every CU streaming at its HBM share, ~10-13 B/clk) because the private L2s do not capture the tiles' operand reuse?
usage (GPU box): cd /tmp && TMPDIR=/tmp python $GRAFT_REPO_ROOT/tools/pmc_config.py <out_file> Bs Bt T D F C [bf16|f32] [xcd_aware 0|2] [phase_tiles csv]"""
import collections, csv, glob, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch
    from ta3n_amd.engine import TrainEngine
    Bs, Bt, T, D, F, C = (int(v) for v in sys.argv[2:8])
    arith, xcd = sys.argv[8], int(sys.argv[9])
    tiles = [int(v) for v in sys.argv[10].split(",")] if len(sys.argv) > 10 and sys.argv[10] else None
    eng = TrainEngine(Bs, Bt, T, D, F, C, bf16=(arith == "bf16"), bf16_store=(arith == "bf16"), xcd_aware=xcd, phase_tiles=tiles)
    eng.X.uniform_(0, 1)
    for v in eng.param_views().values():
        v.normal_(0, 0.02)
    eng.refresh_bf16(x=True, params=True)
    eng.set_hyper([0.75, 0.75, 0.5], 0.003, 1e-3)
    for _ in range(3):
        eng.fused_step(); eng.sgd_step()
    torch.cuda.synchronize()
    sys.exit(0)

out = sys.argv[1]
shape = sys.argv[2:8]
arith = sys.argv[8] if len(sys.argv) > 8 else "bf16"
xcd = sys.argv[9] if len(sys.argv) > 9 else "0"
tiles = sys.argv[10] if len(sys.argv) > 10 else ""
PASSES = ["TCC_HIT_sum TCC_MISS_sum", "FETCH_SIZE", "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum", "SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE"]
table = collections.OrderedDict()
dur = {}
for p in PASSES:
    d = f"/tmp/pmcc_{os.getpid()}_{p.split()[0]}"
    subprocess.run(["rm", "-rf", d])
    cmd = ["rocprofv3", "--pmc", *p.split(), "--kernel-trace", "-d", d, "-o", "out", "--output-format", "csv", "--",
           sys.executable, os.path.abspath(__file__), "--child", *shape, arith, xcd, tiles]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        table[f"pass {p}"] = {"error": r.stdout[-300:]}
        continue
    rows = list(csv.DictReader(open(files[0])))
    by = collections.OrderedDict()
    for r_ in rows:
        by.setdefault((int(r_["Dispatch_Id"]), r_["Kernel_Name"]), {})[r_["Counter_Name"]] = float(r_["Counter_Value"])
    disp = list(by.items())
    sgd = [i for i, ((_, name), _) in enumerate(disp) if "sgd" in name]
    last = disp[sgd[-2] + 1: sgd[-1] + 1] if len(sgd) >= 2 else disp
    for k, ((did, name), v) in enumerate(last):
        table.setdefault(k, {"kernel": name.replace("void ", "").replace("ta3n::", "").replace("(anonymous namespace)::", "")[:48]}).update(v)
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if kt and not dur:
        rows = list(csv.DictReader(open(kt[0])))
        rows.sort(key=lambda r_: int(r_["Start_Timestamp"]))
        sg = [i for i, r_ in enumerate(rows) if "sgd" in r_["Kernel_Name"]]
        lastk = rows[sg[-2] + 1: sg[-1] + 1] if len(sg) >= 2 else rows
        for k, r_ in enumerate(lastk):
            dur[k] = (int(r_["End_Timestamp"]) - int(r_["Start_Timestamp"])) / 1e3
with open(out, "a") as f:
    f.write(f"## shape {' '.join(shape)} {arith} xcd_aware={xcd} tiles={tiles or 'default'} (last of 3 fused steps; durations from the first pass's kernel trace, PMC active)\n")
    for k, v in table.items():
        if "error" in v:
            f.write(f"{k}: {v['error']}\n"); continue
        hit, miss = v.get("TCC_HIT_sum", 0), v.get("TCC_MISS_sum", 0)
        fetch = 2 * 1024 * v.get("FETCH_SIZE", 0)
        us = dur.get(k, 0)
        w = v.get("SQ_WAVES", 0) or 1
        f.write(f"{k} {v['kernel']:48s} {us:8.1f} us  L2 hit {100 * hit / max(hit + miss, 1):5.1f}% ({hit:.3g}/{miss:.3g})  fabric read {fetch / 1e6:8.1f} MB"
                f" = {fetch / max(us, 1e-9) / 1e6:6.2f} TB/s  L1->L2 req {v.get('TCP_TCC_READ_REQ_sum', 0):.3g}  EA rdreq {v.get('TCC_EA0_RDREQ_sum', 0):.3g} (DRAM {v.get('TCC_EA0_RDREQ_DRAM_sum', 0):.3g})"
                f"  LDS conflict/active {v.get('SQ_LDS_BANK_CONFLICT', 0) / max(v.get('SQ_LDS_IDX_ACTIVE', 1), 1):.3f}"
                f"  MFMA busy/w {v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / w:.0f}  wave cyc/w {v.get('SQ_WAVE_CYCLES', 0) / w:.0f}  wait/w {v.get('SQ_WAIT_ANY', 0) / w:.0f}  GUI {v.get('GRBM_GUI_ACTIVE', 0):.0f}\n")
print(open(out).read()[-6000:])
