#!/bin/bash
# round 4, call V: LDS fill rate per CU by loader (LDS-DMA vs register-staged), waves, depth, source (tools/proto_fill.hip)
mkdir -p gpurun_out
timeout 150 tools/proto_fill > gpurun_out/v_proto_fill.txt 2>&1 < /dev/null
echo "rc $?" >> gpurun_out/v_proto_fill.txt
cat gpurun_out/v_proto_fill.txt | tail -n 100
