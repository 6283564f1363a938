#!/bin/bash
# Round 5, GPU call 16: host side of a multi-step call - the ta3n_hyper array filled column-wise and the first step enqueued on its own
# (the rest of the schedule is prepared while the GPU runs it) against the per-entry loop in front of the first launch
# (TA3N_HOST_PREP=legacy).  The whole default tier first (the change is in TrainEngine.train_steps), then the A/B and the round's lines.
set -x
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r05g
mkdir -p $O
cd $R
timeout 480 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; echo "tests rc=$?" >> $O/tests.txt; tail -4 $O/tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
one() { local label="$1"; shift
  python bench.py --single-dtype --no-other-configs --skip-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', round(d['ms_per_step'],4), d.get('ms_per_step_fresh_batch') and round(d['ms_per_step_fresh_batch'],4))" >> $O/host_prep_ab.txt
}
for rep in 1 2 3; do
  one "column-wise + split  cfg2 bf16  20 steps" --steps 20 --warmup 5
  TA3N_HOST_PREP=legacy one "per-entry loop       cfg2 bf16  20 steps" --steps 20 --warmup 5
done
one "column-wise + split  cfg2 bf16 200 steps" --steps 200 --warmup 20
TA3N_HOST_PREP=legacy one "per-entry loop       cfg2 bf16 200 steps" --steps 200 --warmup 20
one "column-wise + split  cfg2 f32   20 steps" --dtype f32 --steps 20 --warmup 5
TA3N_HOST_PREP=legacy one "per-entry loop       cfg2 f32   20 steps" --dtype f32 --steps 20 --warmup 5
for c in 1 4 5; do
  one "column-wise + split  cfg$c       20 steps" --config $c --steps 20 --warmup 5
  TA3N_HOST_PREP=legacy one "per-entry loop       cfg$c       20 steps" --config $c --steps 20 --warmup 5
done
cat $O/host_prep_ab.txt
timeout 200 python bench.py --steps 20 --warmup 5 > $O/bench_driver_protocol.json 2> $O/bench.err; echo "bench rc=$?"
timeout 200 python bench.py --skip-cpu-baseline > $O/bench.json 2>> $O/bench.err
for f in bench_driver_protocol bench; do python -c "
import json; d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$f', d['ms_per_step'], d['value'], 'fresh', d.get('ms_per_step_fresh_batch'), 'frac', round(r['frac'],4), 'traffic_fresh', r['traffic_source'].get('fresh'), 'f32', r['other_arithmetic']['ms_per_step'], {k:round(v['ms_per_step'],4) for k,v in d['configs'].items()}, 'cpu', d.get('cpu_baseline',{}).get('kind'), d.get('cpu_baseline',{}).get('value'))"; done
