#!/bin/bash
# round 4, call M: half-stage parity again; vendor GEMM at configs[3]'s products (with kernel names)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_bf16.py -q -m gpu -k "half_stage or register_blocked" 2>&1 | tail -8 > gpurun_out/m_tests.txt
cat gpurun_out/m_tests.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/m_blaslt_prof -- python $GRAFT_REPO_ROOT/tools/blaslt_calibration.py configs3 > $GRAFT_REPO_ROOT/gpurun_out/m_blaslt_configs3.txt 2>&1
cd $GRAFT_REPO_ROOT
cat gpurun_out/m_blaslt_configs3.txt | tail -20
f=$(ls gpurun_out/m_blaslt_prof/*/*kernel_stats.csv | head -1); head -30 $f | cut -c1-200
