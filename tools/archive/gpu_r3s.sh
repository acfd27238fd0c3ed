#!/bin/bash
# L2 hit rate + wait split of the training attention kernels (generator step, batch 4 and 32)
OUT=$PWD/gpurun_out; REPO=$PWD; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_attn_$i -o pmc -- python $REPO/tools/train_bench.py --batches ${1:-4} --steps 1 > $OUT/pmc_attn_$i.log 2>&1
  echo "pmc $i exit $?"
  cd $REPO; python tools/rocpd_summary.py pmc $(ls $OUT/pmc_attn_$i/*results.db $OUT/pmc_attn_$i/*/*results.db 2>/dev/null | head -1) > $OUT/pmc_attns_$i.txt; cd /tmp
  grep -E '^kernel|fused|^at_fwd|wgrad_partial|db_conv_wgrad|ffn_train_bwd_a' $OUT/pmc_attns_$i.txt | cut -c1-300
  rm -rf $OUT/pmc_attn_$i
done
