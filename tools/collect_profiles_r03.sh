#!/bin/bash
# rocprofv3 evidence of round 3 (GPU box): kernel stats + the kernel trace of the default bench command (gap analysis), PMC passes
# (traffic, instruction mix) on the shipped binary, the default bench line, the data-parallel self-tests (RCCL and the peer
# all-reduce, fp32 / bf16 transport) and the chained-launch A/B.  Results under gpurun_out/prof_r03/ (copy summaries to profiles/).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
rocprofv3 --kernel-trace --stats -d /tmp/kt -o out --output-format csv -- python $R/bench.py --skip-cpu-baseline --steps 20 --warmup 5 > $O/bench_under_rocprof.json 2> /dev/null
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
cp $(find /tmp/kt -name "*kernel_trace.csv" | head -1) /tmp/kt_trace.csv
python $R/tools/trace_gaps.py /tmp/kt_trace.csv 1 --first > $O/gaps_driver_protocol.txt 2>&1
python $R/tools/measure_traffic.py $O > $O/traffic_stdout.txt 2>&1
cp $O/gemm_traffic.json $R/profiles/gemm_traffic.json      # so that the bench line below reports it as fresh
cd $R
python bench.py --steps 20 --warmup 5 > $O/bench_driver_protocol.json 2> $O/bench.err
python bench.py > $O/bench.json 2>> $O/bench.err
for t in fp32 bf16; do
  TA3N_DDP_SELFTEST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 1 --steps 200 --warmup 20 --skip-cpu-baseline --single-dtype --grad-transport $t > $O/ddp_selftest_rccl_$t.json 2>> $O/bench.err
  TA3N_DDP_PEER=1 TA3N_DDP_SELFTEST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 \
    bench.py --gpus 1 --steps 200 --warmup 20 --skip-cpu-baseline --single-dtype --grad-transport $t > $O/ddp_selftest_peer_$t.json 2>> $O/bench.err
done
python tools/time_module_path.py > $O/module_path.txt 2>&1
head -14 $O/bench_kernel_stats.csv | cut -c1-200
cat $O/gaps_driver_protocol.txt | head -14
cat $O/gemm_traffic.json | head -30
for f in bench_driver_protocol bench ddp_selftest_rccl_fp32 ddp_selftest_peer_fp32 ddp_selftest_rccl_bf16 ddp_selftest_peer_bf16; do tail -1 $O/$f.json | cut -c1-260; done
tail -5 $O/module_path.txt
