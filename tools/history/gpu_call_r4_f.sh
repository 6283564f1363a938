#!/bin/bash
# round 4, GPU call F: early refill (lib_ab/, -DTA3N_EARLY_REFILL=1) against the default order (lib/): parity subset on the variant, then A/B bench
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=$PWD/gpurun_out
TA3N_LIBDIR=$PWD/ta3n_amd/lib_ab timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16.py tests/test_gpu_kind_kernels.py tests/test_gpu_pair_twins.py -m gpu -q -x > $O/r4f_tests.txt 2>&1; echo "tests rc $?" >> $O/r4f_tests.txt
grep -E "^FAILED|^ERROR|passed|failed|^E  " $O/r4f_tests.txt | tail -8
summ() { python -c "
import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d['roofline']
print(sys.argv[2], 'bf16', round(1e3*d['ms_per_step'],1), 'f32', round(1e3*r['other_arithmetic']['ms_per_step'],1), 'f32x3', round(1e3*r['split_arithmetic']['ms_per_step'],1), {k: round(1e3*v['ms_per_step'],1) for k, v in d['configs'].items()}, 'phases', [p[3] for p in r['per_phase_us']], 'f32 phases', [p[3] for p in r['other_arithmetic'].get('per_phase_us', [])] if 'per_phase_us' in r['other_arithmetic'] else '')
" $1 $2; }
rm -f $O/r4f_ab.txt
for rep in 1 2; do
  timeout 300 python bench.py --steps 200 --warmup 20 --skip-cpu-baseline > $O/r4f_base_$rep.json 2>>$O/r4f.err; summ $O/r4f_base_$rep.json default >> $O/r4f_ab.txt
  TA3N_LIBDIR=$PWD/ta3n_amd/lib_ab timeout 300 python bench.py --steps 200 --warmup 20 --skip-cpu-baseline > $O/r4f_early_$rep.json 2>>$O/r4f.err; summ $O/r4f_early_$rep.json early-refill >> $O/r4f_ab.txt
done
cat $O/r4f_ab.txt
