#!/bin/bash
# Round 5, GPU call 15: the side jobs of the GEMM launches with 32 / 8 loads in flight per round trip (column sums of the heads kernel's
# per-workgroup partials: 32 - 96 rows; the loss-partial sum) against the previous build (ta3n_amd/lib_t12), alternating; HBM-side traffic
# of the candidate sources; one coordinate-descent sweep of the configs[3] tile list on the final kernels.
set -x
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r05f
mkdir -p $O
cd $R
timeout 120 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/tests_parity.txt 2>&1; echo "parity rc=$?" >> $O/tests_parity.txt; tail -3 $O/tests_parity.txt
PV=$R/ta3n_amd/lib_t12
one() { local label="$1"; shift
  python bench.py --single-dtype --no-other-configs --skip-cpu-baseline --no-fresh-batch "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', round(d['ms_per_step'],4), [p[3] for p in d['roofline']['per_phase_us']])" >> $O/side_batch_ab.txt
}
for rep in 1 2; do
  one "batch 32  cfg2 bf16" --steps 100 --warmup 20
  TA3N_LIBDIR=$PV one "before    cfg2 bf16" --steps 100 --warmup 20
  one "batch 32  cfg2 f32 " --dtype f32 --steps 100 --warmup 20
  TA3N_LIBDIR=$PV one "before    cfg2 f32 " --dtype f32 --steps 100 --warmup 20
  one "batch 32  cfg4     " --config 4 --steps 40 --warmup 10
  TA3N_LIBDIR=$PV one "before    cfg4     " --config 4 --steps 40 --warmup 10
  one "batch 32  cfg5     " --config 5 --steps 40 --warmup 10
  TA3N_LIBDIR=$PV one "before    cfg5     " --config 5 --steps 40 --warmup 10
  one "batch 32  cfg1     " --config 1 --steps 100 --warmup 20
  TA3N_LIBDIR=$PV one "before    cfg1     " --config 1 --steps 100 --warmup 20
done
cat $O/side_batch_ab.txt
cd /tmp && export TMPDIR=/tmp
timeout 200 python $R/tools/measure_traffic.py $O > $O/traffic_stdout.txt 2>&1
head -5 $O/gemm_traffic.json
cd $R
TA3N_TUNE_LAUNCHES=14,11,15,13,12 TA3N_TUNE_CANDS=2222,3222,12222,13222,22222,23222,32222,32221,35221,36222,6222 TA3N_TUNE_SHAPE=512,512,9,2048,512,30 timeout 170 python tools/tune_in_sequence.py bf16 1 > $O/tune_config4_bf16.txt 2>&1; grep -v amdgpu $O/tune_config4_bf16.txt | tail -14
