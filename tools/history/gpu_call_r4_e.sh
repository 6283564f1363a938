#!/bin/bash
# round 4, GPU call E: DMA pieces interleaved with the MFMAs (lib/) against the round-3 order (lib_ab/, -DTA3N_DMA_INTERLEAVE=0): parity subset, then A/B bench
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=$PWD/gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16.py tests/test_gpu_kind_kernels.py tests/test_gpu_pair_twins.py tests/test_gpu_split_k.py tests/test_gpu_chain.py tests/test_gpu_training_equivalence.py "tests/test_main_dropin.py::test_own_main_fused_fast_path_logs_what_the_module_path_logs" -m gpu -q -x > $O/r4e_tests.txt 2>&1; echo "tests rc $?" >> $O/r4e_tests.txt
grep -E "^FAILED|^ERROR|passed|failed|^E  " $O/r4e_tests.txt | tail -12
cat $O/main_fast_vs_module_path.txt 2>/dev/null | cut -c1-230
summ() { python -c "
import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d['roofline']
print(sys.argv[2], 'bf16', round(1e3*d['ms_per_step'],1), 'f32', round(1e3*r['other_arithmetic']['ms_per_step'],1), 'f32x3', round(1e3*r['split_arithmetic']['ms_per_step'],1), {k: round(1e3*v['ms_per_step'],1) for k, v in d['configs'].items()}, 'phases', [p[3] for p in r['per_phase_us']])
" $1 $2; }
for rep in 1 2; do
  timeout 300 python bench.py --steps 200 --warmup 20 --skip-cpu-baseline > $O/r4e_new_$rep.json 2>>$O/r4e.err; summ $O/r4e_new_$rep.json interleaved >> $O/r4e_ab.txt
  TA3N_LIBDIR=$PWD/ta3n_amd/lib_ab timeout 300 python bench.py --steps 200 --warmup 20 --skip-cpu-baseline > $O/r4e_old_$rep.json 2>>$O/r4e.err; summ $O/r4e_old_$rep.json round3-order >> $O/r4e_ab.txt
done
cat $O/r4e_ab.txt
