#!/bin/bash
# round 4, GPU call G: heads kernel with stage A's operands requested before the weight burst (lib/) against the round-3 order (lib_ab/), + the main.py fast-path test
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=$PWD/gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py "tests/test_main_dropin.py::test_own_main_fused_fast_path_logs_what_the_module_path_logs" "tests/test_gpu_gradients.py::test_full_shape_gradients_match_the_oracle" tests/test_gpu_train_steps.py -m gpu -q -x > $O/r4g_tests.txt 2>&1; echo "tests rc $?" >> $O/r4g_tests.txt
grep -E "^FAILED|^ERROR|passed|failed|^E  " $O/r4g_tests.txt | tail -8
summ() { python -c "
import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d['roofline']
print(sys.argv[2], 'bf16', round(1e3*d['ms_per_step'],1), 'f32', round(1e3*r['other_arithmetic']['ms_per_step'],1), {k: round(1e3*v['ms_per_step'],1) for k, v in d['configs'].items()}, 'phases', [p[3] for p in r['per_phase_us']])
" $1 $2; }
rm -f $O/r4g_ab.txt
for rep in 1 2 3; do
  timeout 300 python bench.py --steps 200 --warmup 20 --skip-cpu-baseline > $O/r4g_new_$rep.json 2>>$O/r4g.err; summ $O/r4g_new_$rep.json early-A >> $O/r4g_ab.txt
  TA3N_LIBDIR=$PWD/ta3n_amd/lib_ab timeout 300 python bench.py --steps 200 --warmup 20 --skip-cpu-baseline > $O/r4g_old_$rep.json 2>>$O/r4g.err; summ $O/r4g_old_$rep.json round3-heads >> $O/r4g_ab.txt
done
cat $O/r4g_ab.txt
