#!/bin/bash
# round 4, GPU call Q: the whole GPU suite on the final tree, smoke, then a reduced profile collection (source hash of gemm_traffic.json fresh)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=$PWD/gpurun_out
rm -rf $O/prof_r04
timeout 720 python -m pytest tests -m gpu -q > $O/r4q_tests.txt 2>&1 < /dev/null; echo "tests rc $?" >> $O/r4q_tests.txt
grep -E "^FAILED|^ERROR|passed|failed" $O/r4q_tests.txt | tail -10
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/r4q_smoke.txt 2>&1 < /dev/null; tail -n 2 $O/r4q_smoke.txt
PROTO_RUNS=2 C5_RUNS=3 SKIP_MODULE_PATH=1 timeout 420 bash tools/collect_profiles_r04.sh 2>&1 < /dev/null | tail -n 40
