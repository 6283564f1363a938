#!/bin/bash
# Round 6, GPU call 27: the full -m gpu tier and smoke() on the final tree (host-side changes since call 25: main.py's engine path for DAN / JAN / MCD, bench.py variants).
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6s27; rm -rf $O; mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/gpu_tier.txt 2>&1; echo "gpu tier rc=$? $(tail -1 $O/gpu_tier.txt)" | tee -a $O/summary.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$? $(tail -2 $O/smoke.txt | tr '\n' ' ' | cut -c1-300)" | tee -a $O/summary.txt
