#!/bin/bash
# Round 5, GPU call 1: the re-ordered test tier (new / changed tests first), the new bench line, and two cheap A/Bs
# (persistent workgroups without overlap; fp32 kernel with three LDS stages on the one-tile-per-CU launches).
set -x
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
timeout 900 python -m pytest tests/test_gpu_masked_gradients.py tests/test_gpu_train_steps.py tests/test_gpu_two_stream.py tests/test_gpu_rccl.py \
    tests/test_main_dropin.py tests/test_feature_store.py tests/test_index.py tests/test_gpu_parity.py -m gpu -x -q -s > gpurun_out/r5a_tests.txt 2>&1
echo "tests rc=$?" >> gpurun_out/r5a_tests.txt
tail -5 gpurun_out/r5a_tests.txt
timeout 300 python -m pytest tests/test_gpu_peer.py -m gpu_ab -q > gpurun_out/r5a_peer.txt 2>&1; echo "peer rc=$?" >> gpurun_out/r5a_peer.txt; tail -3 gpurun_out/r5a_peer.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r5a_bench.json 2> gpurun_out/r5a_bench.err; echo "bench rc=$?"
# A/B 1: persistent workgroups (no overlap) - fp32 headline and configs[3]
for P in 0 512 1024; do
  TA3N_PERSIST=$P timeout 200 python bench.py --dtype f32 --single-dtype --no-other-configs --skip-cpu-baseline --no-fresh-batch --steps 100 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('persist $P f32 ms', d['ms_per_step'], [p[3] for p in d['roofline']['per_phase_us']])" >> gpurun_out/r5a_ab.txt
done
for P in 0 256 512; do
  TA3N_PERSIST=$P timeout 200 python bench.py --config 4 --single-dtype --no-other-configs --skip-cpu-baseline --no-fresh-batch --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('persist $P configs[3] ms', d['ms_per_step'], [p[3] for p in d['roofline']['per_phase_us']])" >> gpurun_out/r5a_ab.txt
done
for P in 0 512; do
  TA3N_PERSIST=$P timeout 200 python bench.py --single-dtype --no-other-configs --skip-cpu-baseline --no-fresh-batch --steps 100 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('persist $P bf16 ms', d['ms_per_step'], [p[3] for p in d['roofline']['per_phase_us']])" >> gpurun_out/r5a_ab.txt
done
# A/B 2: fp32 three-stage kernels on the launches with one tile per CU
TA3N_TUNE_COMBOS="124,114,118,124,124,124;3124,114,118,124,124,3124;3124,114,118,124,3124,3124;3124,3114,118,124,124,3124;3124,3114,3118,3124,3124,3124" \
  timeout 300 python tools/tune_in_sequence.py f32 >> gpurun_out/r5a_ab.txt 2>&1
cat gpurun_out/r5a_ab.txt
