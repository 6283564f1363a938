cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmcb
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_WAVES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_MFMA" "SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_LDS"; do
  tag=$(echo $set | cut -d' ' -f2)
  rocprofv3 --pmc $set --kernel-trace -d /tmp/p_$tag -o out --output-format csv -- python $R/tools/prof_step.py 0 0 fused bf16 > /dev/null 2>&1
  f=$(find /tmp/p_$tag -name "*counter_collection.csv" | head -1)
  echo "## pass: $set" >> $R/gpurun_out/pmcb/summary.txt
  python $R/tools/pmc_summary.py $f 200 | tail -9 >> $R/gpurun_out/pmcb/summary.txt
done
cat $R/gpurun_out/pmcb/summary.txt
