#!/bin/bash
# Round 5, GPU call 8: heads kernel after the ISA fixes (tuple ranges by v_readlane, label by scalar load, per-step scalars hoisted, branch-free
# Zr loads) against the previous build (lib_ab = experiments build of commit "two videos per workgroup"); parity gates first; configs[4] tiles.
set -x
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_masked_gradients.py tests/test_gpu_two_stream.py tests/test_gpu_train_steps.py tests/test_gpu_bf16.py -m gpu -x -q -k "not blocked and not half_stage" > gpurun_out/r5h_tests.txt 2>&1
echo "tests rc=$?" >> gpurun_out/r5h_tests.txt; tail -4 gpurun_out/r5h_tests.txt
one() { local label="$1"; shift
  python bench.py --single-dtype --no-other-configs --skip-cpu-baseline --no-fresh-batch "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', round(d['ms_per_step'],4), [p[3] for p in d['roofline']['per_phase_us']])" >> gpurun_out/r5h_ab.txt
}
AB=$PWD/ta3n_amd/lib_ab
for rep in 1 2 3; do
  one "heads isa-fix cfg2" --steps 100 --warmup 20
  TA3N_LIBDIR=$AB one "heads before  cfg2" --steps 100 --warmup 20
  one "heads isa-fix cfg4" --config 4 --steps 40 --warmup 10
  TA3N_LIBDIR=$AB one "heads before  cfg4" --config 4 --steps 40 --warmup 10
  one "heads isa-fix cfg5" --config 5 --steps 40 --warmup 10
  TA3N_LIBDIR=$AB one "heads before  cfg5" --config 5 --steps 40 --warmup 10
done
Z=0,0,0,0,0,0,0,0,0,0
for rep in 1 2; do
  one "cfg5 tiles base      " --config 5 --steps 40 --warmup 10
  one "cfg5 L6=35221        " --config 5 --steps 40 --warmup 10 --phase-tiles $Z,0,0,0,0,35221,0
  one "cfg5 L1=2222         " --config 5 --steps 40 --warmup 10 --phase-tiles $Z,2222,0,0,0,0,0
done
cat gpurun_out/r5h_ab.txt
