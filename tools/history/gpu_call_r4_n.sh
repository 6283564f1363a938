#!/bin/bash
# round 4, call N: 192x128 / 256x128 half-stage tiles - parity, then A/B at configs[3]
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_bf16.py -q -m gpu -x -k "46221 or 56221 or (half_stage and headline)" 2>&1 | tail -15 > gpurun_out/n_tests.txt
cat gpurun_out/n_tests.txt
AB_CANDS=0,32222,7222,46221,56221 timeout 200 python tools/half_stage_ab.py 512 512 9 2048 512 30 > gpurun_out/n_ab_configs3.txt 2>&1
AB_WGRADS_LATE=1 AB_CANDS=0,32222,7222,46221,56221 timeout 200 python tools/half_stage_ab.py 512 512 9 2048 512 30 > gpurun_out/n_ab_configs3_late.txt 2>&1
tail -n 14 gpurun_out/n_ab_configs3.txt; tail -n 14 gpurun_out/n_ab_configs3_late.txt
