#!/bin/bash
# Round 6, GPU call 22: dis_DA DAN / JAN on one rank from the library (ta3n_discrepancy): tests, step times of the DA variants.
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6s22; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_da_extras.py tests/test_gpu_da_over_ranks.py tests/test_gpu_ddp_engine.py tests/test_feature_store.py -m gpu -q -x > $O/tests_da.txt 2>&1; echo "DA tests rc=$? $(tail -1 $O/tests_da.txt)" | tee -a $O/summary.txt
grep -E "^FAILED|Error|assert" $O/tests_da.txt | head -20 | cut -c1-300
python tools/time_da_variants.py 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
TA3N_NATIVE_DISCREPANCY=0 python tools/time_da_variants.py 2>&1 | grep -E "DAN|JAN" | sed 's/^/[torch glue] /' | tee -a $O/summary.txt
