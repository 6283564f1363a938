#!/bin/bash
# round 4, GPU call I: stage timeline of the heads kernel (debug build with cycle stamps) at the three benchmarked shapes
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TA3N_LIBDIR=$PWD/ta3n_amd/lib_ab
for sh in "128 74 5 2048 512 12" "512 512 9 2048 512 30" "128 128 12 1024 512 12"; do
  timeout 120 python tools/heads_timing.py $sh --bf16 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r4i_heads_timing.txt
