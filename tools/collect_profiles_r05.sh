#!/bin/bash
# rocprofv3 evidence of round 5 (GPU box), on the FINAL kernel sources: kernel stats + trace of the bench command under the driver's protocol,
# PMC traffic passes (-> profiles/gemm_traffic.json with the source hash: bench.py then reports roofline.traffic.fresh = true), per-launch
# counters at the headline shape and at configs[3], the bench lines (driver protocol x2, 200 steps).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r05
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
setsid bash -c "rocprofv3 --kernel-trace --stats -d /tmp/kt -o out --output-format csv -- python $R/bench.py --skip-cpu-baseline --steps 20 --warmup 5 > $O/bench_under_rocprof.json 2> /dev/null < /dev/null" &
rp=$!; wait $rp; kill -- -$rp 2> /dev/null
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
cp $(find /tmp/kt -name "*kernel_trace.csv" | head -1) /tmp/kt_trace.csv
python $R/tools/trace_gaps.py /tmp/kt_trace.csv 1 --first > $O/gaps_driver_protocol.txt 2>&1
python $R/tools/measure_traffic.py $O > $O/traffic_stdout.txt 2>&1
cp $O/gemm_traffic.json $R/profiles/gemm_traffic.json      # so that the bench lines below report it as fresh
rm -f $O/pmc_per_launch.txt
timeout 400 python $R/tools/pmc_config.py $O/pmc_per_launch.txt 128 74 5 2048 512 12 bf16 0 > /dev/null 2>&1
timeout 400 python $R/tools/pmc_config.py $O/pmc_per_launch.txt 512 512 9 2048 512 30 bf16 0 0,0,0,0,0,0,0,0,0,0,35221,32222,2222,2222,32222,3222 > /dev/null 2>&1
cd $R
for i in 1 2; do python bench.py --steps 20 --warmup 5 > $O/bench_driver_protocol_$i.json 2>> $O/bench.err; done
python bench.py > $O/bench.json 2>> $O/bench.err
head -14 $O/bench_kernel_stats.csv | cut -c1-200
cat $O/gaps_driver_protocol.txt | head -14
cat $O/gemm_traffic.json
for f in bench_driver_protocol_1 bench_driver_protocol_2 bench; do python -c "
import json,sys; d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$f', d['ms_per_step'], d['value'], 'fresh', d.get('ms_per_step_fresh_batch'), 'frac', round(r['frac'],4), 'traffic_fresh', r['traffic_source'].get('fresh'), 'f32', r['other_arithmetic']['ms_per_step'], round(r['other_arithmetic']['frac'],4), {k:round(v['ms_per_step'],4) for k,v in d['configs'].items()}, 'cpu', d.get('cpu_baseline',{}).get('kind'), d.get('cpu_baseline',{}).get('value'))"; done
