#!/bin/bash
# Round 5, GPU call 7: the whole default tier on the final tree (driver's command), heads stage timeline (cycle stamps) at three shapes on
# the round-5 kernel, the bench line with the bf16-packed store in the fresh-batch loop.
set -x
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r5g_tests.txt 2>&1
echo "tests rc=$?" >> gpurun_out/r5g_tests.txt
tail -4 gpurun_out/r5g_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5g_smoke.txt 2>&1; tail -3 gpurun_out/r5g_smoke.txt
for shape in "128 74 5 2048 512 12" "512 512 9 2048 512 30" "128 128 12 1024 512 12"; do
  python tools/heads_timing.py $shape --bf16 >> gpurun_out/r5g_heads_timeline.txt 2>&1
done
cat gpurun_out/r5g_heads_timeline.txt | grep -v amdgpu
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r5g_bench.json 2> gpurun_out/r5g_bench.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r5g_bench.json'))
print('bf16', d['ms_per_step'], 'fresh', d.get('ms_per_step_fresh_batch'), d.get('fresh_batch_error'), 'f32', d['other_arithmetic']['ms_per_step'], d['other_arithmetic'].get('ms_per_step_fresh_batch'), {k:v.get('ms_per_step') for k,v in d['configs'].items()})
print(d['roofline']['traffic_source'].get('fresh'), d['cpu_baseline']['kind'], d['cpu_baseline']['value'])"
