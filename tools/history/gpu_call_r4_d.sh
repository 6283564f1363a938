#!/bin/bash
# round 4, GPU call D: the tests that failed in call C, then per-launch cache counters at configs[3] (XCD-aware order on / off) and at the headline shape
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=$PWD/gpurun_out
timeout 900 python -m pytest tests/test_gpu_peer.py tests/test_gpu_training_equivalence.py tests/test_main_dropin.py "tests/test_gpu_gradients.py::test_bf16_distance_from_the_fp32_reference_logits_and_gradients" -m gpu -q > $O/r4d_tests.txt 2>&1; echo "tests rc $?" >> $O/r4d_tests.txt
grep -E "^FAILED|^ERROR|passed|failed|^E  " $O/r4d_tests.txt | tail -20
rm -f $O/r4d_pmc.txt
cd /tmp && export TMPDIR=/tmp
timeout 400 python $GRAFT_REPO_ROOT/tools/pmc_config.py $O/r4d_pmc.txt 512 512 9 2048 512 30 bf16 0 > /dev/null 2>$O/r4d_pmc.err
timeout 400 python $GRAFT_REPO_ROOT/tools/pmc_config.py $O/r4d_pmc.txt 512 512 9 2048 512 30 bf16 2 > /dev/null 2>>$O/r4d_pmc.err
timeout 400 python $GRAFT_REPO_ROOT/tools/pmc_config.py $O/r4d_pmc.txt 128 74 5 2048 512 12 bf16 0 > /dev/null 2>>$O/r4d_pmc.err
cat $O/r4d_pmc.txt | cut -c1-420
