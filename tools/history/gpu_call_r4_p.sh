#!/bin/bash
# round 4, call P: configs[3]'s shared-FC launch on 7222 vs 32222 under bench.py's protocol, alternating processes
mkdir -p gpurun_out
out=gpurun_out/p_ab.txt; : > $out
run() { python bench.py --config "$1" --steps 20 --warmup 5 --skip-cpu-baseline --single-dtype 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', d['ms_per_step'], d['value'])" >> $out; }
for i in 1 2 3; do
  TA3N_PHASE_TILES=10:32222 run 4 "configs3 shared-FC 32222"
  run 4 "configs3 shared-FC 7222"
  TA3N_PHASE_TILES=10:2222 run 4 "configs3 shared-FC 2222"
done
cat $out
