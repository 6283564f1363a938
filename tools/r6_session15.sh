#!/bin/bash
# Round 6, GPU call 15: BatchNorm launches that keep their column slab in registers (one pass over memory instead of three) - every test that touches use_bn,
# then the bench line (variants: headline + AdaBN).
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6s15; rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_engine_bn.py tests/test_gpu_engine_avgpool_da.py tests/test_gpu_da_extras.py tests/test_gpu_da_over_ranks.py tests/test_gpu_ddp_engine.py tests/test_main_dropin.py tests/test_train_ddp.py tests/test_gpu_module.py -m gpu -x -q > $O/tests.txt 2>&1; echo "use_bn-related tests rc=$? $(grep -E 'passed|failed' $O/tests.txt | tail -1)" | tee -a $O/summary.txt
grep -E "^E  " $O/tests.txt | head -10 | cut -c1-300 | tee -a $O/summary.txt
python bench.py --steps 20 --warmup 5 > $O/bench.json 2>> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
python -c "
import json; d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1])
print(round(d['ms_per_step'],4), d.get('variants'))" | tee -a $O/summary.txt
