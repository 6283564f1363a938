#!/bin/bash
# Round 6, GPU call 23: ens_DA MCD's loss assembly from the library (ta3n_mcd_source_loss / ta3n_mcd_second_loss): tests, step times of the DA variants.
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6s23; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_engine_mcd.py tests/test_gpu_da_extras.py tests/test_gpu_da_over_ranks.py tests/test_feature_store.py -m gpu -q -x > $O/tests_mcd.txt 2>&1; echo "MCD tests rc=$? $(tail -1 $O/tests_mcd.txt)" | tee -a $O/summary.txt
grep -E "^FAILED|^E  " $O/tests_mcd.txt | head -20 | cut -c1-300
python tools/time_da_variants.py 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
TA3N_NATIVE_MCD=0 python tools/time_da_variants.py 2>&1 | grep -E "MCD" | sed 's/^/[torch glue] /' | tee -a $O/summary.txt
