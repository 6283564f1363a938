#!/bin/bash
# rocprofv3 evidence of round 4 (GPU box): kernel stats + kernel trace of the default bench command under the driver's protocol (gap analysis),
# PMC traffic passes on the shipped binary, the bench lines (driver protocol x3, 200 steps), configs[4] under its protocol x5.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r04
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
# (own session + group kill afterwards: a profiler child that outlives its command must not keep the call open)
setsid bash -c "rocprofv3 --kernel-trace --stats -d /tmp/kt -o out --output-format csv -- python $R/bench.py --skip-cpu-baseline --steps 20 --warmup 5 > $O/bench_under_rocprof.json 2> /dev/null < /dev/null" &
rp=$!; wait $rp; kill -- -$rp 2> /dev/null
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
cp $(find /tmp/kt -name "*kernel_trace.csv" | head -1) /tmp/kt_trace.csv
python $R/tools/trace_gaps.py /tmp/kt_trace.csv 1 --first > $O/gaps_driver_protocol.txt 2>&1
python $R/tools/measure_traffic.py $O > $O/traffic_stdout.txt 2>&1
cp $O/gemm_traffic.json $R/profiles/gemm_traffic.json      # so that the bench lines below report it as fresh
cd $R
for i in $(seq 1 ${PROTO_RUNS:-3}); do python bench.py --steps 20 --warmup 5 > $O/bench_driver_protocol_$i.json 2>> $O/bench.err; done
python bench.py > $O/bench.json 2>> $O/bench.err
for i in $(seq 1 ${C5_RUNS:-5}); do python bench.py --config 5 --steps 20 --warmup 5 --skip-cpu-baseline --single-dtype 2>>$O/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config']['launch'])" >> $O/config5_protocol.txt; done
[ "${SKIP_MODULE_PATH:-0}" = 1 ] || python tools/time_module_path.py > $O/module_path.txt 2>&1
head -14 $O/bench_kernel_stats.csv | cut -c1-200
cat $O/gaps_driver_protocol.txt | head -14
cat $O/config5_protocol.txt
for f in $(for i in $(seq 1 ${PROTO_RUNS:-3}); do echo bench_driver_protocol_$i; done) bench; do python -c "
import json,sys; d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$f', d['ms_per_step'], d['value'], 'frac', round(r['frac'],4), 'f32', r['other_arithmetic']['ms_per_step'], r['other_arithmetic']['frac'], {k:round(v['ms_per_step'],4) for k,v in d['configs'].items()}, 'cpu', d.get('cpu_baseline',{}).get('value'))"; done
[ -f $O/module_path.txt ] && tail -5 $O/module_path.txt
