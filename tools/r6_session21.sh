#!/bin/bash
# Round 6, GPU call 21: the batch assembly of a pipelined step inside the launch that opens it (sgd_open_feed_kernel): the tests that feed steps from a store, the bench lines.
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6s21; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_train_steps.py tests/test_feature_store.py tests/test_gpu_pair_twins.py tests/test_gpu_fused_update.py tests/test_index.py tests/test_gpu_two_stream.py -m gpu -q -x > $O/tests_feed.txt 2>&1; echo "feed tests rc=$? $(tail -1 $O/tests_feed.txt)" | tee -a $O/summary.txt
for i in 1 2 3; do timeout 600 python bench.py --steps 20 --warmup 5 --skip-cpu-baseline > $O/bench_$i.json 2>> $O/bench.err; python -c "
import json; d=json.loads([l for l in open('$O/bench_$i.json') if l.startswith('{')][-1])
print('bench $i', round(d['ms_per_step'],4), 'fresh', round(d['ms_per_step_fresh_batch'],4), 'adabn', round(d['variants']['headline+AdaBN']['ms_per_step'],4))" | tee -a $O/summary.txt; done
