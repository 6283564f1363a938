#!/bin/bash
# Round 6, GPU call 19: BatchNorm launches with row-contiguous 16-byte accesses (4 columns x 1 024 row lanes per workgroup): the tests that run them, per-launch times, the variant's bench figure.
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6s19; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_engine_bn.py -m gpu -q -x > $O/tests_bn.txt 2>&1; echo "bn tests rc=$? $(tail -1 $O/tests_bn.txt)" | tee -a $O/summary.txt
for a in bf16 f32; do timeout 300 python tools/time_bn_phases.py $a 2>/dev/null | tee -a $O/summary.txt; done
for i in 1; do timeout 600 python bench.py --steps 20 --warmup 5 --skip-cpu-baseline > $O/bench_$i.json 2>> $O/bench.err; python -c "
import json; d=json.loads([l for l in open('$O/bench_$i.json') if l.startswith('{')][-1])
print('bench $i', round(d['ms_per_step'],4), 'adabn', round(d['variants']['headline+AdaBN']['ms_per_step'],4))" | tee -a $O/summary.txt; done
