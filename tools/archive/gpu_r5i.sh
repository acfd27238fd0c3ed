#!/bin/bash
# PMC passes of the bench command (matrix-pipe busy + stall / instruction counters) - look at the ffn32 kernels
OUT=$PWD/gpurun_out; REPO=$PWD
BENCH="$REPO/bench.py --no-cpu-baseline --no-f32 --no-f16x1 --no-train --no-extra"
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_WAVE_CYCLES" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_ffn_$i -o pmc -- python $BENCH --steps 1 --warmup 1 > $OUT/pmc_ffn_$i.log 2>&1; echo "pmc $i $?"
done
cd $REPO
db() { ls $1/*results.db $1/*/*results.db 2>/dev/null | head -1; }
python tools/rocpd_summary.py pmc $(db $OUT/pmc_ffn_1) > $OUT/r5i_pmc.txt
python tools/rocpd_summary.py pmc $(db $OUT/pmc_ffn_2) $(db $OUT/pmc_ffn_3) > $OUT/r5i_stall.txt
rm -rf $OUT/pmc_ffn_*
