#!/bin/bash
# Round 5, GPU call 5: the whole default tier on the current build (pipelined heads, two videos per workgroup from 225 videos, tile 35221),
# the experiments tier on the experiments build, four videos per heads workgroup at configs[3] (with its error output this time).
set -x
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r5e_tests.txt 2>&1
echo "tests rc=$?" >> gpurun_out/r5e_tests.txt
tail -6 gpurun_out/r5e_tests.txt
TA3N_LIBDIR=$PWD/ta3n_amd/lib_ab timeout 900 python -m pytest tests -m gpu_ab -q > gpurun_out/r5e_tests_ab.txt 2>&1
echo "ab tests rc=$?" >> gpurun_out/r5e_tests_ab.txt
tail -6 gpurun_out/r5e_tests_ab.txt
for V in 2 4 2 4; do
  TA3N_HEADS_VPW=$V python bench.py --single-dtype --no-other-configs --skip-cpu-baseline --no-fresh-batch --config 4 --steps 40 --warmup 10 2> gpurun_out/r5e_vpw$V.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('vpw=$V cfg4', round(d['ms_per_step'],4), [p[3] for p in d['roofline']['per_phase_us']])" >> gpurun_out/r5e_ab.txt
  tail -3 gpurun_out/r5e_vpw$V.err
done
cat gpurun_out/r5e_ab.txt
