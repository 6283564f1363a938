#!/bin/bash
# round 4, call O: the third-stage rule under configs[4]'s own protocol (two streams, one call), old rule vs new, alternating processes;
# then configs[3] and the headline under the driver's protocol with the new defaults
mkdir -p gpurun_out
out=gpurun_out/o_ab.txt; : > $out
run() { python bench.py --config "$1" --steps 20 --warmup 5 --skip-cpu-baseline --single-dtype 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', d['ms_per_step'], d['value'])" >> $out; }
for i in 1 2 3; do
  TA3N_THIRD_STAGE=always run 5 "configs4 old-rule"
  run 5 "configs4 new-rule"
done
for i in 1 2; do
  TA3N_THIRD_STAGE=always run 4 "configs3 old-rule(+7222)"
  run 4 "configs3 new"
done
run 2 "headline new"; run 2 "headline new"
cat $out
