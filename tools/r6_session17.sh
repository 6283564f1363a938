#!/bin/bash
# Round 6, GPU call 17: the three bench lines once more (cpu_baseline now times every candidate thread count), into the final-evidence directory.
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r06; mkdir -p $O
cd $R
for i in 1 2; do python bench.py --steps 20 --warmup 5 > $O/bench_driver_protocol_$i.json 2>> $O/bench.err; echo "bench protocol $i rc=$?"; done
python bench.py > $O/bench.json 2>> $O/bench.err; echo "bench default rc=$?"
for f in bench_driver_protocol_1 bench_driver_protocol_2 bench; do python -c "
import json; d=json.loads([l for l in open('$O/$f.json') if l.startswith('{')][-1]); c=d['cpu_baseline']
print('$f', round(d['ms_per_step'],4), round(d.get('ms_per_step_fresh_batch',0),4), d['roofline']['traffic_source'].get('fresh'), 'cpu', round(c['value']), c['cores'], c['probe_ms_per_step_by_threads'], 'adabn', round(d['variants']['headline+AdaBN']['ms_per_step'],4))"; done
