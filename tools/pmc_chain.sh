#!/bin/bash
# PMC counters of the forward / backward chains (separate passes), summaries -> gpurun_out/pmc_chain/
R=$PWD; OUT=$R/gpurun_out/pmc_chain; rm -rf $OUT; mkdir -p $OUT
MODE=${1:-f16s}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR" "SQ_INSTS_MFMA SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_SENDMSG" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pc_$i -o p -- python $R/tools/run_kernels_once.py $MODE > $OUT/log_$i.txt 2>&1
  f=$(find /tmp/pc_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/tools/pmc_table.py $f >> $OUT/pmc.txt || echo "set $i failed: $set" >> $OUT/pmc.txt
done
cat $OUT/pmc.txt
