#!/bin/bash
# Round 6, GPU call 2: use_bn inside the fused step (GPU tests); xcd_aware 3 (affinity-group tile order) against the default order at
# configs[3] / headline bf16 / headline fp32 under bench.py's protocol, alternating, with per-launch times; per-launch L2 / fabric
# counters of configs[3] under both orders; workgroup stamps of the fp32 and bf16 launches (where the K loop's time goes).
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6s2; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_engine_bn.py tests/test_gpu_parity.py -m gpu -x -q > $O/tests_bn_parity.txt 2>&1; echo "bn + parity tests rc=$? $(tail -1 $O/tests_bn_parity.txt)" | tee -a $O/summary.txt
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[2], round(d["ms_per_step"], 4), [p[3] for p in d["roofline"]["per_phase_us"]])
except Exception as ex:
    print(sys.argv[2], "unreadable", ex)
PY
}
for rep in 1 2; do
  for x in 0 3; do
    python bench.py --config 4 --xcd $x --single-dtype --no-other-configs --skip-cpu-baseline --no-fresh-batch --steps 40 --warmup 10 > $O/c4_x$x.$rep.json 2>> $O/bench.err
    line $O/c4_x$x.$rep.json "configs[3] bf16 xcd=$x rep $rep" | tee -a $O/summary.txt
  done
done
for rep in 1 2; do
  for x in 0 3; do
    for dt in bf16 f32; do
      python bench.py --config 2 --dtype $dt --xcd $x --single-dtype --no-other-configs --skip-cpu-baseline --no-fresh-batch --steps 100 --warmup 20 > $O/c2_${dt}_x$x.$rep.json 2>> $O/bench.err
      line $O/c2_${dt}_x$x.$rep.json "headline $dt xcd=$x rep $rep" | tee -a $O/summary.txt
    done
  done
done
for x in 0 3; do
  python bench.py --config 5 --xcd $x --single-dtype --no-other-configs --skip-cpu-baseline --no-fresh-batch --steps 40 --warmup 10 > $O/c5_x$x.json 2>> $O/bench.err
  line $O/c5_x$x.json "configs[4] bf16 xcd=$x" | tee -a $O/summary.txt
done
( cd /tmp && export TMPDIR=/tmp
  for x in 0 3; do
    timeout 400 python $R/tools/pmc_config.py $O/pmc_per_launch.txt 512 512 9 2048 512 30 bf16 $x > /dev/null 2>&1
  done
  timeout 400 python $R/tools/pmc_config.py $O/pmc_per_launch.txt 128 74 5 2048 512 12 f32 0 > /dev/null 2>&1 )
for a in "f32 2" "bf16 2" "bf16 4"; do
  TA3N_LIBDIR=$R/ta3n_amd/lib_stamps timeout 200 python tools/gemm_stamps.py $a >> $O/gemm_stamps.txt 2>&1
done
cat $O/pmc_per_launch.txt | cut -c1-260 | tee -a $O/summary.txt
cat $O/gemm_stamps.txt | grep -v "^$" | cut -c1-300 | tee -a $O/summary.txt
timeout 120 tools/proto_fill middle > $O/proto_fused_middle_floor.txt 2>&1; cat $O/proto_fused_middle_floor.txt | tee -a $O/summary.txt
for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout 600 python -m pytest tests/test_gpu_peer.py -m gpu -x -q > $O/peer.$i.txt 2>&1
  echo "peer tests (IPC export retried in the library) rep $i rc=$? $(grep -E 'passed|failed|skipped' $O/peer.$i.txt | tail -1)" | tee -a $O/summary.txt
done
