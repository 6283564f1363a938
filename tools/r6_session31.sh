#!/bin/bash
# Round 6, GPU call 31: does the way the host waits for the GPU matter under the 20-step protocol?  HSA_ENABLE_INTERRUPT=0 (ROCr polls its completion signals instead of sleeping on an interrupt), alternating with the default, bench.py --steps 20 --warmup 5.
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6s31; rm -rf $O; mkdir -p $O
cd $R
for rep in 1 2 3 4; do for v in default poll; do
  if [ $v = poll ]; then export HSA_ENABLE_INTERRUPT=0; else unset HSA_ENABLE_INTERRUPT; fi
  python bench.py --steps 20 --warmup 5 --skip-cpu-baseline --no-other-configs --single-dtype > $O/b.json 2>> $O/bench.err
  python -c "
import json; d=json.loads([l for l in open('$O/b.json') if l.startswith('{')][-1])
print('$v rep $rep: bf16', round(d['ms_per_step'],4), 'fresh', round(d.get('ms_per_step_fresh_batch',0),4))" | tee -a $O/summary.txt
done; done
