#!/bin/bash
# Round 6, GPU call 9: main.py's fused fast path with use_bn AdaBN against the module path (train.log lines, checkpoints), and the other drop-in tests.
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6s9; rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_main_dropin.py tests/test_train_ddp.py -m gpu -x -q > $O/tests.txt 2>&1; echo "main / train_ddp drop-in tests rc=$? $(grep -E 'passed|failed' $O/tests.txt | tail -1)" | tee -a $O/summary.txt
grep -E "^E  " $O/tests.txt | head -10 | cut -c1-300 | tee -a $O/summary.txt
cp gpurun_out/main_fast_vs_module_path_AdaBN.txt $O/ 2>/dev/null; head -8 $O/main_fast_vs_module_path_AdaBN.txt | cut -c1-260 | tee -a $O/summary.txt
