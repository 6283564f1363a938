#!/bin/bash
# rocprofv3 evidence for the default (bf16) bench run; writes under gpurun_out/prof_bf16/ (copy the summaries to profiles/)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_bf16
mkdir -p $O
rocprofv3 --kernel-trace --stats -d /tmp/kt -o out --output-format csv -- python $R/bench.py --skip-cpu-baseline --single-dtype > $O/bench_under_rocprof.json 2> /dev/null
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace -d /tmp/p_$tag -o out --output-format csv -- python $R/tools/prof_step.py 0 0 fused bf16 > /dev/null 2>&1
  f=$(find /tmp/p_$tag -name "*counter_collection.csv" | head -1)
  echo "## pass: $set" >> $O/pmc.txt
  python $R/tools/pmc_summary.py $f 200 | tail -9 >> $O/pmc.txt
done
head -12 $O/bench_kernel_stats.csv | cut -c1-200
cat $O/pmc.txt
