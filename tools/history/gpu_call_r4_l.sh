#!/bin/bash
# round 4, call L: half-stage kernels - parity, then A/B
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_bf16.py -x -q -m gpu -k "half_stage or register_blocked" 2>&1 | tail -15 > gpurun_out/l_tests.txt
cat gpurun_out/l_tests.txt
timeout 300 python tools/half_stage_ab.py 512 512 9 2048 512 30 > gpurun_out/l_ab_configs3.txt 2>&1
timeout 300 python tools/half_stage_ab.py 128 128 12 1024 512 12 > gpurun_out/l_ab_configs4.txt 2>&1
timeout 300 python tools/half_stage_ab.py 128 74 5 2048 512 12 > gpurun_out/l_ab_headline.txt 2>&1
tail -12 gpurun_out/l_ab_configs3.txt gpurun_out/l_ab_configs4.txt gpurun_out/l_ab_headline.txt
