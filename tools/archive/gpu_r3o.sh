#!/bin/bash
# stall / instruction-mix counters of the training attention kernels (generator step, batch 4)
OUT=$PWD/gpurun_out; REPO=$PWD; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_attn_$i -o pmc -- python $REPO/tools/train_bench.py --batches 4 --steps 1 > $OUT/pmc_attn_$i.log 2>&1
  echo "pmc $i exit $?"
done
cd $REPO
python tools/rocpd_summary.py pmc $(ls $OUT/pmc_attn_1/*results.db $OUT/pmc_attn_1/*/*results.db 2>/dev/null | head -1) > $OUT/pmc_attn_1.txt
python tools/rocpd_summary.py pmc $(ls $OUT/pmc_attn_2/*results.db $OUT/pmc_attn_2/*/*results.db 2>/dev/null | head -1) > $OUT/pmc_attn_2.txt
grep -E '^kernel|^at_|fused' $OUT/pmc_attn_1.txt | cut -c1-260
grep -E '^kernel|^at_|fused' $OUT/pmc_attn_2.txt | cut -c1-260
