#!/bin/bash
# stall / LDS / cache counters of the fused attention backward (generator step, batch 4)
OUT=$PWD/gpurun_out; REPO=$PWD; rm -rf $OUT/pmc_attn_* $OUT/prof_train_b32; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM GRBM_GUI_ACTIVE" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_attn_$i -o pmc -- python $REPO/tools/train_bench.py --batches 4 --steps 1 > $OUT/pmc_attn_$i.log 2>&1
  echo "pmc $i exit $?"
  cd $REPO; python tools/rocpd_summary.py pmc $(ls $OUT/pmc_attn_$i/*results.db $OUT/pmc_attn_$i/*/*results.db 2>/dev/null | head -1) > $OUT/pmc_attnq_$i.txt; cd /tmp
  grep -E '^kernel|fused|^at_fwd' $OUT/pmc_attnq_$i.txt | cut -c1-300
  rm -rf $OUT/pmc_attn_$i
done
