#!/bin/bash
# Round 6, GPU call 10: heads kernel with stages B (class logits) and C (video-discriminator hidden layer) as ONE basic block (no barrier between them, B branch-free) -
# parity tests on the new library, then A/B against the previous library (ta3n_amd/lib_prev) under bench.py protocol, alternating.
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6s10; rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_gradients.py tests/test_gpu_masked_gradients.py tests/test_gpu_bf16.py tests/test_gpu_train_steps.py tests/test_gpu_two_stream.py tests/test_gpu_engine_bn.py -m gpu -x -q > $O/tests.txt 2>&1; echo "parity / gradients / bf16 / train_steps / two-stream / bn tests rc=$? $(tail -1 $O/tests.txt)" | tee -a $O/summary.txt
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[2], round(d["ms_per_step"], 4), [p[3] for p in d["roofline"]["per_phase_us"]])
except Exception as ex:
    print(sys.argv[2], "unreadable", ex)
PY
}
for rep in 1 2 3; do
  for lib in lib_prev lib; do
    for c in "2 bf16 100 20" "2 f32 100 20" "4 bf16 40 10" "5 bf16 40 10"; do
      set -- $c
      TA3N_ALLOW_STALE_LIB=1 TA3N_LIBDIR=$R/ta3n_amd/$lib python bench.py --config $1 --dtype $2 --single-dtype --no-other-configs --skip-cpu-baseline --no-fresh-batch --steps $3 --warmup $4 > $O/c$1_$2_$lib.$rep.json 2>> $O/bench.err
      line $O/c$1_$2_$lib.$rep.json "config $1 $2 $lib rep $rep" | tee -a $O/summary.txt
    done
  done
done
