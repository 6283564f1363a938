#!/bin/bash
# round 4, GPU call A: new tests (two-stream one-call step, the reference's own main.py on the GPU), the default bench line,
# configs[4] under the driver's 20-step protocol, 5 runs each with the tuned tiles and with the plan's heuristic
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_two_stream.py tests/test_main_dropin.py -m gpu -x -q > $O/r4a_tests.txt 2>&1; echo "tests rc $?" >> $O/r4a_tests.txt
tail -5 $O/r4a_tests.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/r4a_bench_default.json 2> $O/r4a_bench_default.err; echo "bench rc $?"
for i in 1 2 3 4 5; do
  timeout 300 python bench.py --config 5 --steps 20 --warmup 5 --skip-cpu-baseline --single-dtype 2>>$O/r4a_c5.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tuned', d['ms_per_step'], d['config']['phase_tiles'], d['config']['launch'])" >> $O/r4a_c5.txt
  timeout 300 python bench.py --config 5 --steps 20 --warmup 5 --skip-cpu-baseline --single-dtype --plan-heuristic 2>>$O/r4a_c5.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('heuristic', d['ms_per_step'], d['config']['phase_tiles'])" >> $O/r4a_c5.txt
  timeout 300 python bench.py --config 5 --steps 20 --warmup 5 --skip-cpu-baseline --single-dtype --per-step-calls 2>>$O/r4a_c5.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tuned-per-step-calls', d['ms_per_step'])" >> $O/r4a_c5.txt
done
cat $O/r4a_c5.txt
python -c "
import json; d=json.load(open('$O/r4a_bench_default.json'))
print(d['ms_per_step'], d['value']); print({k:(v.get('ms_per_step'),) for k,v in d.get('configs',{}).items()}); print(d['roofline']['other_arithmetic']['ms_per_step'])"
