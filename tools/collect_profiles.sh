#!/bin/bash
# rocprofv3 evidence of the round: kernel stats of the default bench command, PMC passes (traffic, MFMA / VALU mix), bench line.
# usage (GPU box): bash tools/collect_profiles.sh ; results under gpurun_out/prof_r02/ (copy the summaries to profiles/)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
rocprofv3 --kernel-trace --stats -d /tmp/kt -o out --output-format csv -- python $R/bench.py --skip-cpu-baseline > $O/bench_under_rocprof.json 2> /dev/null
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
python $R/tools/measure_traffic.py $O > $O/traffic_stdout.txt 2>&1
python $R/bench.py > $O/bench.json 2> $O/bench.err
head -14 $O/bench_kernel_stats.csv | cut -c1-220
cat $O/gemm_traffic.json
