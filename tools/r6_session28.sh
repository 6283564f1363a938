#!/bin/bash
# Round 6, GPU call 28: rocprofv3 --kernel-trace --stats of the DA variants' train loops (headline shape, bf16 twins, 50 steps each): the kernels behind bench.py's `variants`.
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6s28; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in adabn mcd dan jan; do
  rm -rf /tmp/kt_$v
  setsid bash -c "rocprofv3 --kernel-trace --stats -d /tmp/kt_$v -o out --output-format csv -- python $R/tools/da_variant_loop.py $v 50 > /tmp/loop_$v.txt 2>&1 < /dev/null" &
  rp=$!; wait $rp; kill -- -$rp 2> /dev/null
  f=$(find /tmp/kt_$v -name "*kernel_stats.csv" | head -1)
  echo "## $v (50 train steps; $(tail -1 /tmp/loop_$v.txt))" >> $O/da_variants_kernel_stats.txt
  python - "$f" >> $O/da_variants_kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# kernel time per step {tot / 50 / 1e3:.1f} us over {sum(int(r['Calls']) for r in rows) / 50:.1f} launches")
for r in rows[:16]:
    print(f"{r['Name'][:90]:90s} calls/step {int(r['Calls']) / 50:5.1f}  avg {float(r['AverageNs']) / 1e3:7.2f} us  {float(r['Percentage']):5.1f} %")
PY
done
tail -80 $O/da_variants_kernel_stats.txt
