#!/bin/bash
# Round 6, GPU call 33: ens_DA MCD's two passes on twins as well (the engine copies the parameter / input twins into the second workspace): tests, step times.
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6s33; rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_engine_mcd.py tests/test_gpu_da_extras.py tests/test_gpu_da_over_ranks.py tests/test_feature_store.py tests/test_main_dropin.py tests/test_gpu_pair_twins.py -m gpu -q -x > $O/tests.txt 2>&1; echo "tests rc=$? $(tail -1 $O/tests.txt)" | tee -a $O/summary.txt
grep -E "^FAILED|^E  " $O/tests.txt | head -12 | cut -c1-300
python tools/time_da_variants.py 2>&1 | grep -v amdgpu.ids | grep bf16 | tee -a $O/summary.txt
