#!/bin/bash
# Round 5, GPU call 11: batched partial sums (gradient-norm partials in the optimiser launches and side workgroups, COLSUM side tasks: eight
# loads in flight per round trip instead of one) against the previous build (lib_prev); parity / step tests first.
set -x
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_train_steps.py tests/test_gpu_two_stream.py tests/test_gpu_avgpool.py tests/test_gpu_rccl.py -m gpu -x -q > gpurun_out/r5k_tests.txt 2>&1
echo "tests rc=$?" >> gpurun_out/r5k_tests.txt; tail -3 gpurun_out/r5k_tests.txt
one() { local label="$1"; shift
  python bench.py --single-dtype --no-other-configs --skip-cpu-baseline --no-fresh-batch "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', round(d['ms_per_step'],4), [p[3] for p in d['roofline']['per_phase_us']])" >> gpurun_out/r5k_ab.txt
}
PV=$PWD/ta3n_amd/lib_prev
for rep in 1 2 3; do
  one "batched sums cfg2 bf16" --steps 100 --warmup 20
  TA3N_LIBDIR=$PV one "before       cfg2 bf16" --steps 100 --warmup 20
  one "batched sums cfg2 f32 " --dtype f32 --steps 100 --warmup 20
  TA3N_LIBDIR=$PV one "before       cfg2 f32 " --dtype f32 --steps 100 --warmup 20
  one "batched sums cfg4     " --config 4 --steps 40 --warmup 10
  TA3N_LIBDIR=$PV one "before       cfg4     " --config 4 --steps 40 --warmup 10
done
cat gpurun_out/r5k_ab.txt
