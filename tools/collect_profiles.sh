#!/bin/bash
# rocprofv3 evidence of the round: kernel stats of the default bench command, PMC passes (traffic, MFMA / VALU mix), the bench
# lines of every BASELINE configuration, the 1-rank RCCL self-test, the bf16 parity-gate report.
# usage (GPU box): bash tools/collect_profiles.sh ; results under gpurun_out/prof_r02/ (copy the summaries to profiles/)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
rocprofv3 --kernel-trace --stats -d /tmp/kt -o out --output-format csv -- python $R/bench.py --skip-cpu-baseline > $O/bench_under_rocprof.json 2> /dev/null
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
python $R/tools/measure_traffic.py $O > $O/traffic_stdout.txt 2>&1
cp $O/gemm_traffic.json $R/profiles/gemm_traffic.json      # so that the bench line below reports it as fresh
# (also writes pmc_bf16.txt / pmc_f32.txt / pmc_f32x3.txt: per-dispatch counter tables of the last fused step per arithmetic)
cd $R
python bench.py > $O/bench.json 2> $O/bench.err
for c in 1 4 5; do python bench.py --config $c --steps 50 --warmup 10 > $O/bench_config$c.json 2>> $O/bench.err; done
TA3N_DDP_SELFTEST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --gpus 1 --steps 200 --warmup 20 --skip-cpu-baseline --single-dtype > $O/ddp_selftest_1rank.json 2>> $O/bench.err
python -m pytest tests/test_gpu_bf16.py -q -s -k "oracle" 2>&1 | grep -E "bf16 vs bf16-oracle|passed|failed" > $O/bf16_parity_gate.txt
python tools/time_module_path.py > $O/module_path.txt 2>&1
head -14 $O/bench_kernel_stats.csv | cut -c1-220
cat $O/gemm_traffic.json
tail -3 $O/bf16_parity_gate.txt | cut -c1-300
for f in bench bench_config1 bench_config4 bench_config5 ddp_selftest_1rank; do tail -1 $O/$f.json | cut -c1-330; done
tail -5 $O/module_path.txt
