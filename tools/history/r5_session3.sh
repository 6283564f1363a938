#!/bin/bash
# Round 5, GPU call 3: (a) heads kernel with relation operands requested one relation ahead (lib) against round 4's (lib_ab, experiments
# build of the previous commit); (b) the four-wave 128x128 tile on two half stages (35221: two workgroups per CU) on the long launches of
# configs[3]; parity gates for both first.
set -x
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_two_stream.py tests/test_gpu_train_steps.py "tests/test_gpu_bf16.py" -m gpu -x -q -k "not oracle_gate_at_the_other or 35221" > gpurun_out/r5c_tests.txt 2>&1
echo "tests rc=$?" >> gpurun_out/r5c_tests.txt; tail -4 gpurun_out/r5c_tests.txt
one() { # label, env, args...
  local label="$1"; shift
  python bench.py --single-dtype --no-other-configs --skip-cpu-baseline --no-fresh-batch "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', round(d['ms_per_step'],4), [p[3] for p in d['roofline']['per_phase_us']])" >> gpurun_out/r5c_ab.txt
}
AB=$PWD/ta3n_amd/lib_ab
for rep in 1 2; do
  one "heads new cfg4" --config 4 --steps 40 --warmup 10
  TA3N_LIBDIR=$AB one "heads old cfg4" --config 4 --steps 40 --warmup 10
  one "heads new cfg5" --config 5 --steps 40 --warmup 10
  TA3N_LIBDIR=$AB one "heads old cfg5" --config 5 --steps 40 --warmup 10
  one "heads new cfg2" --steps 100 --warmup 20
  TA3N_LIBDIR=$AB one "heads old cfg2" --steps 100 --warmup 20
done
Z=0,0,0,0,0,0,0,0,0,0
for rep in 1 2; do
  one "tiles base            " --config 4 --steps 40 --warmup 10 --phase-tiles $Z,32222,32222,2222,2222,32222,3222
  one "tiles L1=35221        " --config 4 --steps 40 --warmup 10 --phase-tiles $Z,35221,32222,2222,2222,32222,3222
  one "tiles L2=35221        " --config 4 --steps 40 --warmup 10 --phase-tiles $Z,32222,35221,2222,2222,32222,3222
  one "tiles L6=35221        " --config 4 --steps 40 --warmup 10 --phase-tiles $Z,32222,32222,2222,2222,35221,3222
  one "tiles L1,L2,L6=35221  " --config 4 --steps 40 --warmup 10 --phase-tiles $Z,35221,35221,2222,2222,35221,3222
  one "tiles L1=36222        " --config 4 --steps 40 --warmup 10 --phase-tiles $Z,36222,32222,2222,2222,32222,3222
done
cat gpurun_out/r5c_ab.txt
