#!/bin/bash
# Round 6, GPU call 24: step times of the DA variants once more (best of three runs each: the first run of a configuration pays one-time costs), native and torch glue.
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6s24; rm -rf $O; mkdir -p $O
cd $R
python tools/time_da_variants.py 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
TA3N_NATIVE_MCD=0 TA3N_NATIVE_DISCREPANCY=0 python tools/time_da_variants.py 2>&1 | grep -E "MCD|DAN|JAN" | sed 's/^/[torch glue] /' | tee -a $O/summary.txt
