#!/bin/bash
# Round 6, GPU call 14: the full -m gpu tier exactly as the driver runs it (-x -q), three times in a row on the shipped tree, then smoke().
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6s14; rm -rf $O; mkdir -p $O
cd $R
for i in 1 2 3; do
  python -m pytest tests/ -x -q -m gpu > $O/tier.$i.txt 2>&1; echo "tier run $i rc=$? $(grep -E 'passed|failed' $O/tier.$i.txt | tail -1)" | tee -a $O/summary.txt
done
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$? $(tail -2 $O/smoke.txt | tr '\n' ' ' | cut -c1-300)" | tee -a $O/summary.txt
