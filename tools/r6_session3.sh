#!/bin/bash
# Round 6, GPU call 3: split-K of the shared-FC product (ta3n_config.split_k = 4: every tile of the step's first launch computed by two
# workgroups over the two halves of K, so the launch has two resident workgroups per compute unit instead of one) - experiments build:
# its tests, then A/B against the unsplit launch under bench.py's protocol in fp32 and bf16, alternating.
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6s3; rm -rf $O; mkdir -p $O
cd $R
export TA3N_LIBDIR=$R/ta3n_amd/lib_ab
timeout 900 python -m pytest tests/test_gpu_split_k.py -m gpu_ab -x -q > $O/tests_split_k.txt 2>&1; echo "split-K tests (experiments build) rc=$? $(tail -1 $O/tests_split_k.txt)" | tee -a $O/summary.txt
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
  for sk in 0 4; do
    for dt in f32 bf16; do
      TA3N_SPLIT_K=$sk python bench.py --config 2 --dtype $dt --single-dtype --no-other-configs --skip-cpu-baseline --no-fresh-batch --steps 100 --warmup 20 > $O/c2_${dt}_sk$sk.$rep.json 2>> $O/bench.err
      line $O/c2_${dt}_sk$sk.$rep.json "headline $dt split_k=$sk rep $rep" | tee -a $O/summary.txt
    done
  done
done
