#!/bin/bash
# Round 6, GPU call 29: number of side workgroups that carry the rest of the optimiser update in the step's first GEMM launch (TA3N_SIDE_WGS; shipped: 256) - headline bf16 and fp32, 100 steps, alternating.
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6s29; rm -rf $O; mkdir -p $O
cd $R
for rep in 1 2 3; do for n in 256 128 512 1024; do
  TA3N_SIDE_WGS=$n python bench.py --steps 100 --warmup 20 --skip-cpu-baseline --no-other-configs > $O/b.json 2>> $O/bench.err
  python -c "
import json; d=json.loads([l for l in open('$O/b.json') if l.startswith('{')][-1]); r=d['roofline']
print('side_wgs $n rep $rep: bf16', round(d['ms_per_step'],4), 'fresh', round(d.get('ms_per_step_fresh_batch',0),4), 'f32', round((d.get('other_arithmetic') or {}).get('ms_per_step',0),4))" | tee -a $O/summary.txt
done; done
