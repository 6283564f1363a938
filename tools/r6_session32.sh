#!/bin/bash
# Round 6, GPU call 32: the unfused launch lists read bf16 twins too (a launch family of their own in add_bf16_twins): the tests that run unfused lists in bf16 / f32x3, the DA variants' step times with and without.
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6s32; rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_pair_twins.py tests/test_gpu_gradients.py tests/test_gpu_parity.py tests/test_gpu_da_extras.py tests/test_gpu_engine_mcd.py tests/test_gpu_da_over_ranks.py tests/test_gpu_ddp_engine.py tests/test_feature_store.py tests/test_gpu_training_equivalence.py -m gpu -q -x > $O/tests.txt 2>&1; echo "tests rc=$? $(tail -1 $O/tests.txt)" | tee -a $O/summary.txt
grep -E "^FAILED|^E  " $O/tests.txt | head -12 | cut -c1-300
python tools/time_da_variants.py 2>&1 | grep -v amdgpu.ids | grep bf16 | tee -a $O/summary.txt
TA3N_UNFUSED_TWINS=0 python tools/time_da_variants.py 2>&1 | grep -v amdgpu.ids | grep bf16 | sed 's/^/[no unfused twins] /' | tee -a $O/summary.txt
