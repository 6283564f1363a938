#!/bin/bash
# Round 6, GPU call 5: the two-rank bench path on hardware (two ranks sharing the GPU over gloo: launcher, probe, line) 3 x, the new PIPE parity
# shapes (> 448 videos at 12 / 9 segments), the hardened peer tests 6 x.
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6s5; rm -rf $O; mkdir -p $O
cd $R
for i in 1 2 3; do
  timeout 1000 python -m pytest tests/test_gpu_bench_two_ranks.py -m gpu -x -q > $O/two_ranks.$i.txt 2>&1
  echo "two-rank bench test rep $i rc=$? $(grep -E 'passed|failed|skipped' $O/two_ranks.$i.txt | tail -1)" | tee -a $O/summary.txt
done
TA3N_BENCH_SHARED_GPU=1 TA3N_PEER_TIMEOUT_S=10 python bench.py --gpus 2 --steps 20 --warmup 5 --skip-cpu-baseline --single-dtype --no-other-configs > $O/bench_two_ranks_shared_gpu.json 2> $O/bench_two_ranks_shared_gpu.err; echo "bench --gpus 2 (shared GPU test mode) rc=$?" | tee -a $O/summary.txt
python - <<'PY' | tee -a gpurun_out/r6s5/summary.txt
import json
try:
    d = json.loads([l for l in open("gpurun_out/r6s5/bench_two_ranks_shared_gpu.json") if l.startswith("{")][-1])
    print(d["n_gpus"], d["ms_per_step"], d["config"]["exchange"], json.dumps(d["config"]["exchange_probe"]["candidates"])[:1200])
except Exception as ex:
    print("unreadable", ex)
PY
timeout 1500 python -m pytest tests/test_gpu_gradients.py -m gpu -x -q -k "pipe_T" > $O/pipe_shapes.txt 2>&1; echo "PIPE parity shapes rc=$? $(tail -1 $O/pipe_shapes.txt)" | tee -a $O/summary.txt
grep "pipe_T" $O/pipe_shapes.txt | cut -c1-400 | tee -a $O/summary.txt
for i in 1 2 3 4 5 6; do
  timeout 900 python -m pytest tests/test_gpu_peer.py -m gpu -x -q > $O/peer.$i.txt 2>&1
  echo "peer tests rep $i rc=$? $(grep -E 'passed|failed|skipped' $O/peer.$i.txt | tail -1)" | tee -a $O/summary.txt
done
