#!/bin/bash
# round 4, GPU call J: heads kernel with 2 / 4 videos per workgroup - parity with the variants forced on the golden shapes, the full-shape gradient tests
# (plan's own choice: 4 at configs[3], 2 at configs[4]), then A/B against one video per workgroup
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=$PWD/gpurun_out
for v in 4 2; do
  TA3N_HEADS_VPW=$v timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16.py -m gpu -q -x > $O/r4j_tests_vpw$v.txt 2>&1; echo "vpw $v rc $?"; grep -E "^FAILED|passed|failed|^E  " $O/r4j_tests_vpw$v.txt | tail -4
done
timeout 600 python -m pytest "tests/test_gpu_gradients.py::test_full_shape_gradients_match_the_oracle" "tests/test_gpu_gradients.py::test_bf16_distance_from_the_fp32_reference_logits_and_gradients" tests/test_gpu_two_stream.py -m gpu -q -x > $O/r4j_tests_full.txt 2>&1; echo "full rc $?"; grep -E "^FAILED|passed|failed|^E  " $O/r4j_tests_full.txt | tail -4
rm -f $O/r4j_ab.txt
for rep in 1 2 3; do
  for v in default 1; do
    for c in 4 5; do
      E=""; [ $v = 1 ] && E="TA3N_HEADS_VPW=1"
      env $E timeout 200 python bench.py --config $c --steps 100 --warmup 10 --skip-cpu-baseline --single-dtype 2>>$O/r4j.err | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('vpw=$v config $c', round(1e3 * d['ms_per_step'], 1), 'us; heads', [p[3] for p in r['per_phase_us'] if p[0] == 6])" >> $O/r4j_ab.txt
    done
  done
done
cat $O/r4j_ab.txt
