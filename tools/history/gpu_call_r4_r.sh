#!/bin/bash
# round 4, call R: DA options over ranks (one-GPU emulation) + the engine DA tests that share the refactored code
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_da_over_ranks.py tests/test_gpu_engine_mcd.py tests/test_gpu_da_extras.py tests/test_gpu_engine_bn.py tests/test_gpu_engine_avgpool_da.py -q -m gpu 2>&1 < /dev/null | tail -n 25 > gpurun_out/r_tests.txt
cat gpurun_out/r_tests.txt
