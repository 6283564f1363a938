#!/bin/bash
# round-3 starting point: the driver's protocol (20 steps, 5 warmup) three times, a longer run, and a kernel trace of the
# driver protocol for the gap analysis (tools/trace_gaps.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3a
mkdir -p $O
cd $R
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --skip-cpu-baseline --single-dtype 2>>$O/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('driver-protocol', d['ms_per_step'], d['roofline']['all_kernels_us'])" ; done | tee $O/driver_protocol.txt
python bench.py --steps 200 --warmup 20 --skip-cpu-baseline --single-dtype 2>>$O/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('200-step', d['ms_per_step'], d['roofline']['all_kernels_us'], d['roofline']['per_phase_us'])" | tee -a $O/driver_protocol.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
rocprofv3 --kernel-trace -d /tmp/kt -o out --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --skip-cpu-baseline --single-dtype > $O/bench_under_rocprof.json 2>/dev/null
cp $(find /tmp/kt -name "*kernel_trace.csv" | head -1) $O/kernel_trace_driver_protocol.csv
python $R/tools/trace_gaps.py $O/kernel_trace_driver_protocol.csv 1 | tee $O/gaps_driver_protocol.txt
