#!/bin/bash
# round 4, call W: the clean-built library of the final tree - smoke and the fp32 parity file
mkdir -p gpurun_out
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/w_smoke.txt 2>&1 < /dev/null; tail -n 2 gpurun_out/w_smoke.txt
timeout 200 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 < /dev/null | tail -n 4 > gpurun_out/w_tests.txt; cat gpurun_out/w_tests.txt
