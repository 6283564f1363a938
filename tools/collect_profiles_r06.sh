#!/bin/bash
# rocprofv3 evidence of round 6 (GPU box), on the FINAL kernel sources: the full -m gpu tier, kernel stats + trace of the bench command under
# the driver's protocol, PMC traffic passes (-> profiles/gemm_traffic.json with the source hash, headline arithmetics + configs[3]: bench.py then
# reports roofline.traffic fresh and the per-configuration ratio), per-launch counters, the bench lines (driver protocol x2, 200 steps), the N > 1
# code path in a 1-rank group with the exchange probe.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r06
rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
python -m pytest tests -m gpu -x -q > $O/gpu_tier.txt 2>&1; echo "gpu tier rc=$? $(tail -1 $O/gpu_tier.txt)" | tee -a $O/summary.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
setsid bash -c "rocprofv3 --kernel-trace --stats -d /tmp/kt -o out --output-format csv -- python $R/bench.py --skip-cpu-baseline --steps 20 --warmup 5 > $O/bench_under_rocprof.json 2> /dev/null < /dev/null" &
rp=$!; wait $rp; kill -- -$rp 2> /dev/null
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
cp $(find /tmp/kt -name "*kernel_trace.csv" | head -1) /tmp/kt_trace.csv
python $R/tools/trace_gaps.py /tmp/kt_trace.csv 1 --first > $O/gaps_driver_protocol.txt 2>&1
timeout 900 python $R/tools/measure_traffic.py $O > $O/traffic_stdout.txt 2>&1
cp $O/gemm_traffic.json $R/profiles/gemm_traffic.json      # so that the bench lines below report it as fresh
rm -f $O/pmc_per_launch.txt
timeout 400 python $R/tools/pmc_config.py $O/pmc_per_launch.txt 128 74 5 2048 512 12 bf16 0 > /dev/null 2>&1
timeout 400 python $R/tools/pmc_config.py $O/pmc_per_launch.txt 512 512 9 2048 512 30 bf16 0 > /dev/null 2>&1
cd $R
for i in 1 2; do python bench.py --steps 20 --warmup 5 > $O/bench_driver_protocol_$i.json 2>> $O/bench.err; done
python bench.py > $O/bench.json 2>> $O/bench.err
TA3N_DDP_SELFTEST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29612 python bench.py --steps 100 --warmup 10 --skip-cpu-baseline --single-dtype --no-other-configs > $O/bench_selftest.json 2>> $O/bench.err
TA3N_BENCH_SHARED_GPU=1 TA3N_PEER_TIMEOUT_S=10 python bench.py --gpus 2 --steps 20 --warmup 5 --skip-cpu-baseline --single-dtype --no-other-configs > $O/bench_two_ranks_shared_gpu.json 2> $O/bench_two_ranks_shared_gpu.err; echo "bench --gpus 2, two ranks sharing the GPU over gloo (test mode) rc=$?; stdout lines: $(grep -c . $O/bench_two_ranks_shared_gpu.json)" | tee -a $O/summary.txt
python bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_gpus2.out 2> $O/bench_gpus2.err; echo "bench --gpus 2 on this 1-GPU box rc=$? (must be non-zero): $(tail -1 $O/bench_gpus2.err)" | tee -a $O/summary.txt
head -14 $O/bench_kernel_stats.csv | cut -c1-200 | tee -a $O/summary.txt
head -14 $O/gaps_driver_protocol.txt | tee -a $O/summary.txt
cat $O/gemm_traffic.json | tee -a $O/summary.txt
for f in bench_driver_protocol_1 bench_driver_protocol_2 bench bench_selftest; do python -c "
import json,sys; d=json.loads([l for l in open('$O/$f.json') if l.startswith('{')][-1]); r=d['roofline']
print('$f', d['ms_per_step'], d['value'], 'fresh', d.get('ms_per_step_fresh_batch'), 'frac', round(r['frac'],4), 'traffic', r.get('traffic'), 'traffic_fresh', r['traffic_source'].get('fresh'), 'f32', (r.get('other_arithmetic') or {}).get('ms_per_step'), {k:(round(v['ms_per_step'],4), round(v.get('traffic_over_algorithmic', 0), 2)) for k,v in (d.get('configs') or {}).items()}, 'cpu', (d.get('cpu_baseline') or {}).get('kind'), (d.get('cpu_baseline') or {}).get('value'), (d.get('cpu_baseline') or {}).get('probe_ms_per_step_by_threads'), 'exchange', d['config'].get('exchange'))" 2>&1 | tee -a $O/summary.txt; done
tail -30 $O/pmc_per_launch.txt | cut -c1-250 >> $O/summary.txt
