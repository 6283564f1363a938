#!/bin/bash
# A/B of the chained-launch hand-off protocol knobs (ta3n_gemm.hip: launch_gemm) at the headline shape
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { env "$@" python bench.py --steps 100 --warmup 20 --skip-cpu-baseline --single-dtype 2>&1 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(1e3*d['ms_per_step'],1), [p[3] for p in d['roofline']['per_phase_us']])"; }
run TA3N_CHAIN=0
run TA3N_CHAIN=1 TA3N_CHAIN_SLEEP=4 TA3N_CHAIN_MEMSET=1
run TA3N_CHAIN=1 TA3N_CHAIN_SLEEP=4 TA3N_CHAIN_MEMSET=1 TA3N_CHAIN_RMWPOLL=1
run TA3N_CHAIN=1 TA3N_CHAIN_SLEEP=1 TA3N_CHAIN_MEMSET=1 TA3N_CHAIN_RMWPOLL=1
run TA3N_CHAIN=1 TA3N_CHAIN_SLEEP=4 TA3N_CHAIN_MEMSET=1 TA3N_CHAIN_NOWAIT=1
