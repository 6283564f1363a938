#!/bin/bash
# round 4, GPU call C: the whole GPU suite (no -x), the 1-rank RCCL self-test of the all-reduce and the sharded update
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out
timeout 1700 python -m pytest tests -m gpu -q > $O/r4c_tests.txt 2>&1; echo "tests rc $?" >> $O/r4c_tests.txt
grep -E "^FAILED|^ERROR|passed|failed" $O/r4c_tests.txt | tail -30
B="python -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nproc-per-node 1 bench.py --gpus 1 --steps 200 --warmup 20 --single-dtype --skip-cpu-baseline --no-other-configs"
for v in allreduce sharded2 sharded1; do
  for rep in 1 2 3; do
    case $v in
      allreduce) E="TA3N_DDP_SHARDED=0" ;;
      sharded2) E="TA3N_DDP_SHARDED=1 TA3N_DDP_SHARDED_STREAMS=2" ;;
      sharded1) E="TA3N_DDP_SHARDED=1 TA3N_DDP_SHARDED_STREAMS=1" ;;
    esac
    env TA3N_DDP_SELFTEST=1 $E timeout 300 $B 2>>$O/r4c_selftest.err | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); c = d['config']
        print('$v', round(1e3 * d['ms_per_step'], 2), 'us/step', 'exposed', round(c['collective']['exposed_us_per_step'], 2), 'without', round(1e3 * c['collective']['step_without_collective_ms'], 2), c['gradient_exchange'][:60], 'ranks', c.get('rccl_ranks'), c['launch'][:40])" >> $O/r4c_selftest.txt
  done
done
cat $O/r4c_selftest.txt
