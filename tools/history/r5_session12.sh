#!/bin/bash
# Round 5, GPU call 12 (final tree): the whole default tier as the driver runs it, smoke, PMC traffic on the final sources, bench lines.
set -x
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r5l_tests.txt 2>&1
echo "tests rc=$?" >> gpurun_out/r5l_tests.txt
tail -4 gpurun_out/r5l_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5l_smoke.txt 2>&1; tail -2 gpurun_out/r5l_smoke.txt
R=$PWD; O=$R/gpurun_out/prof_r05c; mkdir -p $O
( cd /tmp && export TMPDIR=/tmp && python $R/tools/measure_traffic.py $O > $O/traffic_stdout.txt 2>&1
  rm -rf /tmp/kt; setsid bash -c "rocprofv3 --kernel-trace --stats -d /tmp/kt -o out --output-format csv -- python $R/bench.py --skip-cpu-baseline --steps 20 --warmup 5 > $O/bench_under_rocprof.json 2> /dev/null < /dev/null" & rp=$!; wait $rp; kill -- -$rp 2> /dev/null
  cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv; cp $(find /tmp/kt -name "*kernel_trace.csv" | head -1) /tmp/kt_trace.csv
  python $R/tools/trace_gaps.py /tmp/kt_trace.csv 1 --first > $O/gaps_driver_protocol.txt 2>&1 )
cp $O/gemm_traffic.json $R/profiles/gemm_traffic.json
for i in 1 2; do python bench.py --steps 20 --warmup 5 > $O/bench_driver_protocol_$i.json 2>> $O/bench.err; done
python bench.py > $O/bench.json 2>> $O/bench.err
head -14 $O/gaps_driver_protocol.txt
head -5 $O/gemm_traffic.json
for f in bench_driver_protocol_1 bench_driver_protocol_2 bench; do python -c "
import json,sys; d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$f', d['ms_per_step'], d['value'], 'fresh', d.get('ms_per_step_fresh_batch'), 'frac', round(r['frac'],4), 'traffic_fresh', r['traffic_source'].get('fresh'), 'f32', r['other_arithmetic']['ms_per_step'], round(r['other_arithmetic']['frac'],4), 'x3', r['split_arithmetic']['ms_per_step'], {k:round(v['ms_per_step'],4) for k,v in d['configs'].items()}, 'cpu', d.get('cpu_baseline',{}).get('kind'), d.get('cpu_baseline',{}).get('value'))"; done
