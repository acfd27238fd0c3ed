#!/bin/bash
# rocprofv3 PMC passes of the adversarial training step (tools/train_bench.py, 32 clips, one step after warm-up): matrix-pipe
# busy, instruction mix, fabric requests and L2 hit / miss per kernel -> gpurun_out/train_pmc_<tag>.txt
TAG=${1:-r06}
OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
         "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $C --kernel-trace -d $OUT/pmct_${TAG}_$i -o pmc -- python $REPO/tools/train_bench.py --batches 32 --steps 1 --adversarial > $OUT/pmct_${TAG}_$i.log 2>&1; echo "pmc $i $?"
done
cd $REPO
db() { ls $1/*results.db $1/*/*results.db 2>/dev/null | head -1; }
python tools/rocpd_summary.py pmc $(db $OUT/pmct_${TAG}_1) $(db $OUT/pmct_${TAG}_2) $(db $OUT/pmct_${TAG}_3) $(db $OUT/pmct_${TAG}_4) $(db $OUT/pmct_${TAG}_5) > $OUT/train_pmc_$TAG.txt
head -40 $OUT/train_pmc_$TAG.txt | cut -c1-400
rm -rf $OUT/pmct_${TAG}_*
