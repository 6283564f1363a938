#!/bin/bash
# Round 6, GPU call 18: the experiments tier (-m gpu_ab) on the experiments build of the FINAL sources.
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6s18; rm -rf $O; mkdir -p $O
cd $R
TA3N_LIBDIR=$R/ta3n_amd/lib_ab timeout 1500 python -m pytest tests -m gpu_ab -q > $O/tests_gpu_ab.txt 2>&1; echo "gpu_ab tier (experiments build) rc=$? $(grep -E 'passed|failed' $O/tests_gpu_ab.txt | tail -1)" | tee -a $O/summary.txt
grep -E "^FAILED" $O/tests_gpu_ab.txt | cut -c1-200 | tee -a $O/summary.txt
