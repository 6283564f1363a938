#!/bin/bash
# round 4, GPU call B: the whole GPU suite (new: training equivalence, contract-relative bf16 bound, peer poison, main.py fast path, reference main.py), smoke, default bench
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > $O/r4b_tests.txt 2>&1; echo "tests rc $?" >> $O/r4b_tests.txt
tail -15 $O/r4b_tests.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/r4b_smoke.txt 2>&1; tail -2 $O/r4b_smoke.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/r4b_bench_default.json 2> $O/r4b_bench_default.err; echo "bench rc $?"
python -c "
import json; d=json.load(open('$O/r4b_bench_default.json'))
print(d['ms_per_step'], d['value']); print({k:v.get('ms_per_step') for k,v in d.get('configs',{}).items()}); print(d['roofline']['other_arithmetic']['ms_per_step']); print(d['cpu_baseline']['value'], d['cpu_baseline']['reference_over_port'])"
