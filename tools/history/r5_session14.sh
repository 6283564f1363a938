#!/bin/bash
# Round 5, GPU call 14 (last): on the shipped tree (library byte-identical to the one of call 12's green tier) - the experiments tier on the
# experiments build of the same sources, the N > 1 code path in a 1-rank RCCL group, per-launch counters and the heads stage timeline on the
# final kernels, kernel stats of the configs[3] bench command, one more line under the driver's protocol.
set -x
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r05e
mkdir -p $O
cd $R
TA3N_LIBDIR=$R/ta3n_amd/lib_ab timeout 400 python -m pytest tests -m gpu_ab -x -q > $O/tests_gpu_ab.txt 2>&1; echo "gpu_ab rc=$?" >> $O/tests_gpu_ab.txt; tail -3 $O/tests_gpu_ab.txt
for t in fp32 bf16; do
  TA3N_DDP_SELFTEST=1 timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 1 --steps 200 --warmup 20 --skip-cpu-baseline --single-dtype --no-other-configs --grad-transport $t > $O/ddp_selftest_rccl_$t.json 2>> $O/bench.err
  python -c "
import json; d=json.loads(open('$O/ddp_selftest_rccl_$t.json').read().strip().splitlines()[-1]); c=d['config']
print('selftest $t', d['ms_per_step'], c.get('gradient_exchange'), c.get('collective'))"
done
cd /tmp && export TMPDIR=/tmp
rm -f $O/pmc_per_launch.txt
timeout 300 python $R/tools/pmc_config.py $O/pmc_per_launch.txt 128 74 5 2048 512 12 bf16 0 > /dev/null 2>&1
timeout 300 python $R/tools/pmc_config.py $O/pmc_per_launch.txt 512 512 9 2048 512 30 bf16 0 0,0,0,0,0,0,0,0,0,0,35221,32222,2222,2222,32222,3222 > /dev/null 2>&1
rm -rf /tmp/kt4
setsid bash -c "timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt4 -o out --output-format csv -- python $R/bench.py --config 4 --single-dtype --no-other-configs --skip-cpu-baseline --no-fresh-batch --steps 20 --warmup 5 > $O/bench_config4_under_rocprof.json 2> /dev/null < /dev/null" &
rp=$!; wait $rp; kill -- -$rp 2> /dev/null
cp $(find /tmp/kt4 -name "*kernel_stats.csv" | head -1) $O/bench_config4_kernel_stats.csv
cp $(find /tmp/kt4 -name "*kernel_trace.csv" | head -1) /tmp/kt4_trace.csv
python $R/tools/trace_gaps.py /tmp/kt4_trace.csv 1 --first > $O/gaps_config4.txt 2>&1
cd $R
for shape in "128 74 5 2048 512 12" "512 512 9 2048 512 30" "128 128 12 1024 512 12"; do
  timeout 120 python tools/heads_timing.py $shape --bf16 >> $O/heads_timeline.txt 2>&1
done
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_driver_protocol.json 2>> $O/bench.err; echo "bench rc=$?"
python -c "
import json; d=json.loads(open('$O/bench_driver_protocol.json').read().strip().splitlines()[-1]); r=d['roofline']
print('bench', d['ms_per_step'], d['value'], 'fresh', d.get('ms_per_step_fresh_batch'), 'frac', round(r['frac'],4), 'traffic_fresh', r['traffic_source'].get('fresh'), 'f32', r['other_arithmetic']['ms_per_step'], {k:round(v['ms_per_step'],4) for k,v in d['configs'].items()}, 'cpu', d['cpu_baseline'].get('kind'), d['cpu_baseline'].get('value'))"
head -12 $O/bench_config4_kernel_stats.csv | cut -c1-180
head -14 $O/gaps_config4.txt
grep -v amdgpu $O/heads_timeline.txt | tail -30
tail -40 $O/pmc_per_launch.txt
