#!/bin/bash
# Round 6, GPU call 11: the bench lines once more with the `variants` entry (headline + AdaBN inside the fused step); driver protocol x2 and the default.
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6s11; rm -rf $O; mkdir -p $O
cd $R
for i in 1 2; do python bench.py --steps 20 --warmup 5 > $O/bench_driver_protocol_$i.json 2>> $O/bench.err; echo "bench protocol $i rc=$? stdout lines $(grep -c . $O/bench_driver_protocol_$i.json)" | tee -a $O/summary.txt; done
python bench.py > $O/bench.json 2>> $O/bench.err; echo "bench default rc=$?" | tee -a $O/summary.txt
for f in bench_driver_protocol_1 bench_driver_protocol_2 bench; do python -c "
import json; d=json.loads([l for l in open('$O/$f.json') if l.startswith('{')][-1]); r=d['roofline']
print('$f', round(d['ms_per_step'],4), 'fresh', round(d.get('ms_per_step_fresh_batch',0),4), 'frac', round(r['frac'],4), 'traffic_fresh', r['traffic_source'].get('fresh'), 'f32', round(d['other_arithmetic']['ms_per_step'],4), {k:round(v['ms_per_step'],4) for k,v in d['configs'].items()}, 'variants', {k:(round(v.get('ms_per_step',0),4), v.get('error')) for k,v in d.get('variants',{}).items()}, 'cpu', round(d['cpu_baseline']['value']))" | tee -a $O/summary.txt; done
