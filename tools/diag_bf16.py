"""Diagnostic (GPU): HIP bf16 step vs plan interpreter (fp64) vs bf16 oracle, per gradient tensor, relative L2."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from golden_util import Golden, case_config, step_schedule
from oracle import ta3n_oracle as orc
from plan_interp import Interp
from ta3n_amd import _lib
from ta3n_amd.engine import TrainEngine
from ta3n_amd.synthetic import synth_batch, synth_state
from test_plan_cpu import ALL_FLAGS, make_hyper

name = sys.argv[1]; store = int(sys.argv[2]) if len(sys.argv) > 2 else 1
g = Golden(name); c = case_config(g); T = c["T"]; st = step_schedule(c)[0]
flags = ALL_FLAGS | _lib.FLAG_BF16_MFMA | (_lib.FLAG_BF16_STORE if store else 0)
eng = TrainEngine(c["Bs"], c["Bt"], T, c["D"], c["fc_dim"], c["C"], dropout_i=0.0, dropout_v=0.0, clip=c["clip"], bf16=True, bf16_store=bool(store))
plan = _lib.Plan(c["Bs"], c["Bt"], T, c["D"], c["fc_dim"], c["C"], flags, phase_tiles=[ph["tile"] % 1000 for ph in eng.plan.description["phases"] if ph["kind"] == 0][:16])
it = Interp(plan)
shapes = {n: s for n, _, s, _ in plan.params}
params = synth_state(shapes, seed=c["wseed"], scale=c["wscale"])
eng.load_state(params); it.set_params(params)
xs, xt, ys, yt = synth_batch(c["C"], T, c["D"], c["Bs"], c["Bt"], seed=st["xseed"])
xs[st["n_src"]:] = 0; xt[st["n_tgt"]:] = 0
eng.set_batch(xs.cuda(), xt.cuda(), ys.cuda())
eng.set_hyper([0.75, 0.75, 0.5], 0.003, st["lr"], train=True, valid_source=st["n_src"], valid_target=st["n_tgt"])
eng.fused_step(); torch.cuda.synchronize()
it.X = torch.cat((xs, xt), 0).double().numpy().reshape(-1); it.labels[:c["Bs"]] = ys.numpy()
it.hy = make_hyper(c, st, T, st["lr"]); it.G[:] = 0; it.run_group(4)
cfg = orc.Config(num_class=c["C"], num_segments=T, feature_dim=c["D"], fc_dim=c["fc_dim"], dropout_i=0.0, dropout_v=0.0, arithmetic="bf16", bf16_twins=bool(store))
res = orc.train_step(orc.TrainState(params={k: v.clone() for k, v in params.items()}, lr=st["lr"]), xs, xt, ys, [0.75, 0.75, 0.5], 0.003, cfg, clip=c["clip"], n_src=st["n_src"], n_tgt=st["n_tgt"])
hip = {k: v.cpu().double().numpy() for k, v in eng.param_views(eng.G).items()}
itp = it.get_params(it.G)
def l2(a, b): return float(np.sqrt(((a - b) ** 2).sum() / ((b ** 2).sum() + 1e-30)))
rows = []
for k, w in res["grads"].items():
    w = w.double().numpy()
    rows.append((l2(hip[k], w), l2(hip[k], itp[k].reshape(w.shape)), l2(itp[k].reshape(w.shape), w), k))
rows.sort(reverse=True)
print(f"{name} store={store}:  hip-vs-oracle  hip-vs-interp  interp-vs-oracle")
for r in rows[:8]: print(f"  {r[0]:.2e}  {r[1]:.2e}  {r[2]:.2e}  {r[3]}")
# forward regions
B = c["Bs"] + c["Bt"]
for nm, off, shape in (("F1", it.g.o_F1, (B * T, it.g.F)), ("Zr", it.g.o_Zr, (B, -1)), ("Hr", it.g.o_Hr, (B, -1)), ("gZ", it.g.o_gZ, (B, -1)), ("gZ1", it.g.o_gZ1, (B * T, it.g.F)), ("gHf", it.g.o_gHf, (B * T, it.g.F))):
    n = int(np.prod([B * T, it.g.F])) if shape[1] != -1 else eng.plan.regions[nm][1]
    a = eng.region(nm).cpu().double().numpy().reshape(-1)[:n]; b = it.ws[off:off + n]
    d = np.abs(a - b); print(f"  region {nm}: rel L2 {l2(a, b):.2e}  max {d.max():.2e} (scale {np.abs(b).max():.2e})  n>1e-3*scale: {(d > 1e-3 * np.abs(b).max()).sum()}")
