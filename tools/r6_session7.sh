#!/bin/bash
# Round 6, GPU call 7: the peer transport after "park and take back" (no free + re-export of an exchange buffer any more): its tests 12 x, the
# two-rank bench test 3 x, the DDP-engine / RCCL tests once.
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6s7; rm -rf $O; mkdir -p $O
cd $R
python -m pytest tests -m gpu -q > $O/gpu_tier.txt 2>&1; echo "full gpu tier (no -x) rc=$? $(grep -E 'passed|failed' $O/gpu_tier.txt | tail -1)" | tee -a $O/summary.txt
grep -E "^FAILED|^ERROR" $O/gpu_tier.txt | cut -c1-200 | tee -a $O/summary.txt
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  timeout 900 python -m pytest tests/test_gpu_peer.py -m gpu -x -q -rs > $O/peer.$i.txt 2>&1
  echo "peer tests rep $i rc=$? $(grep -E 'passed|failed|skipped' $O/peer.$i.txt | tail -1) $(grep -c SKIPPED $O/peer.$i.txt) skip lines" | tee -a $O/summary.txt
done
for i in 1 2 3; do
  timeout 1000 python -m pytest tests/test_gpu_bench_two_ranks.py -m gpu -x -q > $O/two_ranks.$i.txt 2>&1
  echo "two-rank bench test rep $i rc=$? $(grep -E 'passed|failed|skipped' $O/two_ranks.$i.txt | tail -1)" | tee -a $O/summary.txt
done
timeout 900 python -m pytest tests/test_gpu_ddp_engine.py tests/test_gpu_rccl.py -m gpu -x -q > $O/ddp.txt 2>&1; echo "ddp engine + rccl tests rc=$? $(tail -1 $O/ddp.txt)" | tee -a $O/summary.txt
TA3N_DDP_SELFTEST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29613 python bench.py --steps 100 --warmup 10 --skip-cpu-baseline --single-dtype --no-other-configs > $O/bench_selftest.json 2> $O/bench_selftest.err; echo "selftest bench (shared communicator) rc=$?" | tee -a $O/summary.txt
python - <<'PY' | tee -a gpurun_out/r6s7/summary.txt
import json
try:
    d = json.loads([l for l in open("gpurun_out/r6s7/bench_selftest.json") if l.startswith("{")][-1])
    print(d["ms_per_step"], d["config"]["exchange"], d["config"]["rccl_ranks"], json.dumps(d["config"]["exchange_probe"]["candidates"])[:900])
except Exception as ex:
    print("unreadable", ex)
PY
