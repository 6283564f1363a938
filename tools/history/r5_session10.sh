#!/bin/bash
# Round 5, GPU call 10 (final tree): the whole default tier as the driver runs it, smoke, heads A/B against the previous kernel, PMC traffic on
# the final sources, bench lines (driver protocol x2, 200 steps).
set -x
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r5j_tests.txt 2>&1
echo "tests rc=$?" >> gpurun_out/r5j_tests.txt
tail -4 gpurun_out/r5j_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5j_smoke.txt 2>&1; tail -2 gpurun_out/r5j_smoke.txt
one() { local label="$1"; shift
  python bench.py --single-dtype --no-other-configs --skip-cpu-baseline --no-fresh-batch "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', round(d['ms_per_step'],4), [p[3] for p in d['roofline']['per_phase_us']])" >> gpurun_out/r5j_ab.txt
}
AB=$PWD/ta3n_amd/lib_ab
for rep in 1 2; do
  one "heads final  cfg2" --steps 100 --warmup 20
  TA3N_LIBDIR=$AB one "heads before cfg2" --steps 100 --warmup 20
  one "heads final  cfg4" --config 4 --steps 40 --warmup 10
  TA3N_LIBDIR=$AB one "heads before cfg4" --config 4 --steps 40 --warmup 10
  one "heads final  cfg5" --config 5 --steps 40 --warmup 10
  TA3N_LIBDIR=$AB one "heads before cfg5" --config 5 --steps 40 --warmup 10
done
cat gpurun_out/r5j_ab.txt
R=$PWD; O=$R/gpurun_out/prof_r05b; mkdir -p $O
( cd /tmp && export TMPDIR=/tmp && python $R/tools/measure_traffic.py $O > $O/traffic_stdout.txt 2>&1 )
cp $O/gemm_traffic.json $R/profiles/gemm_traffic.json
for i in 1 2; do python bench.py --steps 20 --warmup 5 > $O/bench_driver_protocol_$i.json 2>> $O/bench.err; done
python bench.py > $O/bench.json 2>> $O/bench.err
cat $O/gemm_traffic.json | head -12
for f in bench_driver_protocol_1 bench_driver_protocol_2 bench; do python -c "
import json,sys; d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$f', d['ms_per_step'], d['value'], 'fresh', d.get('ms_per_step_fresh_batch'), 'frac', round(r['frac'],4), 'traffic_fresh', r['traffic_source'].get('fresh'), 'f32', r['other_arithmetic']['ms_per_step'], round(r['other_arithmetic']['frac'],4), 'x3', r['split_arithmetic']['ms_per_step'], {k:round(v['ms_per_step'],4) for k,v in d['configs'].items()}, 'cpu', d.get('cpu_baseline',{}).get('kind'), d.get('cpu_baseline',{}).get('value'))"; done
