#!/bin/bash
# round 4, call S: the DA options under two ranks sharing the GPU (gloo over CUDA tensors)
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_ddp_engine.py -q -m gpu -k "da_options" 2>&1 < /dev/null | tail -n 40 > gpurun_out/s_tests.txt
cat gpurun_out/s_tests.txt
