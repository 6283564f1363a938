#!/bin/bash
# round 4, GPU call K: videos per heads workgroup at configs[3] (1 / 2 / 4) and at the headline shape (1 / 2)
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out; mkdir -p $O; rm -f $O/r4k_ab.txt
run() { env TA3N_HEADS_VPW=$1 timeout 200 python bench.py --config $2 --steps 100 --warmup 10 --skip-cpu-baseline --single-dtype --no-other-configs 2>>$O/r4k.err | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('vpw=$1 config $2', round(1e3 * d['ms_per_step'], 1), 'us; heads', [p[3] for p in r['per_phase_us'] if p[0] == 6])" >> $O/r4k_ab.txt; }
for rep in 1 2; do run 1 4; run 2 4; run 4 4; run 1 2; run 2 2; done
cat $O/r4k_ab.txt
