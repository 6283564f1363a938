#!/bin/bash
# Round 5, GPU call 2: the whole default tier in the driver's order (without -x: see every failure), then the headline bench.
set -x
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r5b_tests.txt 2>&1
echo "tests rc=$?" >> gpurun_out/r5b_tests.txt
tail -15 gpurun_out/r5b_tests.txt
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r5b_bench.json 2> gpurun_out/r5b_bench.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r5b_bench.json'))
print('bf16', d['ms_per_step'], 'fresh', d.get('ms_per_step_fresh_batch'), 'f32', d['other_arithmetic']['ms_per_step'], {k:v.get('ms_per_step') for k,v in d['configs'].items()})
print([p[3] for p in d['roofline']['per_phase_us']]); print([p[3] for p in d['other_arithmetic']['roofline']['per_phase_us']])"
