"""Half-stage (64-k LDS stages, gemm_tiles MODE 5) against full-stage twin kernels: isolated time of every GEMM launch of the plan
under each tile code, then the fused step with the plan's own tiles against a per-launch best-of mix (20-step protocol, 5 runs each,
interleaved).  Debug / measurement aid.
usage: python tools/half_stage_ab.py [Bs Bt T D F C]"""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ta3n_amd.engine import TrainEngine

shape = [int(v) for v in sys.argv[1:7]] if len(sys.argv) >= 7 else [512, 512, 9, 2048, 512, 30]
CANDS = [int(v) for v in os.environ.get("AB_CANDS", "0,2222,7222,32222,36222,46221,56221").split(",")]


def code(ph):
    """tile code of a launch of the plan description (ta3n_amd/tuning.py)"""
    rm, rn = ph.get("rm", 1), ph.get("rn", 1)
    blk = 4 if rm == 3 else 5 if rm == 4 else (rm > 1) + 2 * (rn > 1)
    return ph["tile"] % 1000 + 1000 * ((ph["tile"] // 1000) & 15) + 3000 * ph.get("half_stages", 0) + 10000 * blk


def engine(tile=0, phase_tiles=None):
    eng = TrainEngine(*shape, bf16=True, bf16_store=True, tile_config=tile, phase_tiles=phase_tiles,
                      wgrads_late=os.environ.get("AB_WGRADS_LATE", "0") == "1")
    eng.X.uniform_(0, 1)
    for v in eng.param_views().values():
        v.normal_(0, 0.02)
    eng.refresh_bf16(x=True, params=True)
    eng.set_hyper([0.75, 0.75, 0.5], 0.003, 1e-3)
    return eng


def step_us(eng, steps=20, warmup=5, runs=5):
    out = []
    for _ in range(runs):
        for _ in range(warmup):
            eng.fused_step()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(steps):
            eng.fused_step()
        b.record(); torch.cuda.synchronize()
        out.append(1e3 * a.elapsed_time(b) / steps)
    return out


print("shape", shape, "wgrads_late", os.environ.get("AB_WGRADS_LATE", "0"))
table, tiles = {}, {}
for cand in CANDS:
    eng = engine(cand)
    eng.gemm_phase_times(3)
    table[cand] = eng.gemm_phase_times(20)
    tiles[cand] = [code(ph) for ph in eng.plan.description["phases"] if ph["kind"] == 0]
    del eng
n = len(table[0])
for i in range(n):
    print(f"launch {i}: plan {tiles[0][i]:>6} " + "  ".join(f"{tiles[c][i]}:{1e3 * table[c][i]:.1f}" for c in CANDS), flush=True)
best = []
for i in range(n):
    c = min(CANDS, key=lambda c: table[c][i])
    # keep the plan's own tile unless the winner is clearly ahead in isolation (3 %)
    best.append(tiles[c][i] if table[c][i] < 0.97 * table[0][i] else tiles[0][i])
print("plan tiles     ", tiles[0])
print("best-of tiles  ", best)
e0, e1 = engine(0), engine(0, phase_tiles=best[:16])
print("mixed plan got ", [code(ph) for ph in e1.plan.description["phases"] if ph["kind"] == 0])
r0, r1 = [], []
for _ in range(3):
    r0 += step_us(e0, runs=2); r1 += step_us(e1, runs=2)
print("step, plan tiles    us:", " ".join(f"{v:.1f}" for v in r0), " median", f"{statistics.median(r0):.1f}")
print("step, best-of tiles us:", " ".join(f"{v:.1f}" for v in r1), " median", f"{statistics.median(r1):.1f}")
