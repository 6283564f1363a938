"""Three eager train steps at the BASELINE shape for rocprofv3 (kernel trace / PMC passes).
usage: python tools/prof_step.py [tile_config] [xcd_aware] [fused|unfused] [f32|bf16|f32x3] [bench config number: 2 (default), 4 = configs[3]]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ta3n_amd.engine import TrainEngine
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 0
xcd = int(sys.argv[2]) if len(sys.argv) > 2 else 0
phase_tiles = []
if tile == 0:                     # the bench's measured per-launch tile shapes
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    bf16 = len(sys.argv) > 4 and sys.argv[4] == "bf16"
    phase_tiles = None      # engine default: ta3n_amd/tuning.py
bf16 = len(sys.argv) > 4 and sys.argv[4] == "bf16"
split = len(sys.argv) > 4 and sys.argv[4] == "f32x3"
SHAPES = {2: (128, 74, 5, 2048, 512, 12), 4: (512, 512, 9, 2048, 512, 30)}
shape = SHAPES[int(sys.argv[5]) if len(sys.argv) > 5 else 2]
eng = TrainEngine(*shape, tile_config=tile, xcd_aware=xcd, phase_tiles=phase_tiles, bf16=bf16, bf16_store=(bf16 or split),
                  f32_split=split)
eng.X.uniform_(0, 1)
for v in eng.param_views().values(): v.normal_(0, 0.02)
eng.refresh_bf16(x=True, params=True)
eng.set_hyper([0.75,0.75,0.5], 0.003, 1e-3)
fused = (sys.argv[3] != "unfused") if len(sys.argv) > 3 else True
for _ in range(3):
    if fused and eng.plan.has_fused_step:
        eng.fused_step()
    else:
        eng.forward(); eng.loss(); eng.backward()
    eng.sgd_step()
torch.cuda.synchronize()
