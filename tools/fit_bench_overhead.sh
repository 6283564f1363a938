#!/bin/bash
# total time of the timed region for several K at W = 5: T(K) = a + b K separates the bracket's fixed cost from the step time
for k in 10 20 40 80 160; do
  python bench.py --gpus 1 --steps $k --warmup 5 --skip-cpu-baseline --single-dtype 2>/dev/null | tail -1 | K=$k python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); k=int(os.environ['K']); print(k, 'total_us', round(d['ms_per_step']*1e3*k,1), 'per_step', round(d['ms_per_step']*1e3,2))"
done
