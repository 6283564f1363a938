#!/bin/bash
# round 4, call U: the tests that drive train_ddp.py / the data-parallel engine paths, after the DA-over-ranks and logging changes
mkdir -p gpurun_out
timeout 330 python -m pytest tests/test_feature_store.py tests/test_gpu_ddp_engine.py tests/test_gpu_rccl.py tests/test_train_ddp.py -q -m gpu 2>&1 < /dev/null | tail -n 25 > gpurun_out/u_tests.txt
cat gpurun_out/u_tests.txt
