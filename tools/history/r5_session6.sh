#!/bin/bash
# Round 5, GPU call 6: new parity cases; configs[3] with the TRN weight gradients moved to the last launch (and the gradient-at-F1 launch on
# the four-wave tile); then the round's profile collection on the final kernel sources.
set -x
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ragged" > gpurun_out/r5f_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r5f_tests.txt; tail -3 gpurun_out/r5f_tests.txt
one() { local label="$1"; shift
  python bench.py --single-dtype --no-other-configs --skip-cpu-baseline --no-fresh-batch "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', round(d['ms_per_step'],4), [p[3] for p in d['roofline']['per_phase_us']])" >> gpurun_out/r5f_ab.txt
}
Z=0,0,0,0,0,0,0,0,0,0
for rep in 1 2; do
  one "cfg4 base                  " --config 4 --steps 40 --warmup 10
  one "cfg4 wgrads late           " --config 4 --steps 40 --warmup 10 --wgrads-late 1
  one "cfg4 wgrads late, L6=35221 " --config 4 --steps 40 --warmup 10 --wgrads-late 1 --phase-tiles $Z,35221,32222,2222,2222,35221,3222
  one "cfg4 wgrads late, L7=32222 " --config 4 --steps 40 --warmup 10 --wgrads-late 1 --phase-tiles $Z,35221,32222,2222,2222,32222,32222
done
cat gpurun_out/r5f_ab.txt
bash tools/collect_profiles_r05.sh > gpurun_out/r5f_collect.txt 2>&1
tail -30 gpurun_out/r5f_collect.txt
