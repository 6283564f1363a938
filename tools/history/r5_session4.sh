#!/bin/bash
# Round 5, GPU call 4: (a) videos per heads workgroup now that the relation loops are pipelined; (b) re-tune of the headline tile lists
# on the pruned kernels (one coordinate-descent sweep each, bounded).
set -x
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
one() { local label="$1"; shift
  python bench.py --single-dtype --no-other-configs --skip-cpu-baseline --no-fresh-batch "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', round(d['ms_per_step'],4), [p[3] for p in d['roofline']['per_phase_us']])" >> gpurun_out/r5d_ab.txt
}
for rep in 1 2; do
  for V in 1 2 4; do
    TA3N_HEADS_VPW=$V one "vpw=$V cfg4" --config 4 --steps 40 --warmup 10
    TA3N_HEADS_VPW=$V one "vpw=$V cfg5" --config 5 --steps 40 --warmup 10
  done
done
cat gpurun_out/r5d_ab.txt
timeout 330 python tools/tune_in_sequence.py bf16 1 > gpurun_out/r5d_tune_bf16.txt 2>&1; tail -12 gpurun_out/r5d_tune_bf16.txt
timeout 330 python tools/tune_in_sequence.py f32 1 > gpurun_out/r5d_tune_f32.txt 2>&1; tail -12 gpurun_out/r5d_tune_f32.txt
