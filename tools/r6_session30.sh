#!/bin/bash
# Round 6, GPU call 30: the full -m gpu tier twice more on the final tree (flake check of the tier the driver runs at round end), then bench.py under the driver's protocol.
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6s30; rm -rf $O; mkdir -p $O
cd $R
for i in 1 2; do python -m pytest tests -m gpu -x -q > $O/gpu_tier_$i.txt 2>&1; echo "gpu tier run $i rc=$? $(grep -E 'passed|failed' $O/gpu_tier_$i.txt | tail -1)" | tee -a $O/summary.txt; done
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_protocol.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
python -c "
import json; d=json.loads([l for l in open('$O/bench_driver_protocol.json') if l.startswith('{')][-1])
print('bench', round(d['ms_per_step'],4), 'fresh', round(d['ms_per_step_fresh_batch'],4), {k: round(v.get('ms_per_step',-1),4) for k,v in d['variants'].items()}, 'frac', round(d['roofline']['frac'],4), 'cpu', round(d['cpu_baseline']['value']))" | tee -a $O/summary.txt
