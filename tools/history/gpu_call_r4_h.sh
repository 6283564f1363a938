#!/bin/bash
# round 4, GPU call H: the whole GPU suite on the final tree, smoke, then the round's profile collection
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=$PWD/gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > $O/r4h_tests.txt 2>&1; echo "tests rc $?" >> $O/r4h_tests.txt
grep -E "^FAILED|^ERROR|passed|failed" $O/r4h_tests.txt | tail -10
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/r4h_smoke.txt 2>&1; tail -2 $O/r4h_smoke.txt
timeout 900 bash tools/collect_profiles_r04.sh 2>&1 | tail -40
