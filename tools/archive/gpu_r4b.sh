#!/bin/bash
# round 4, session b: where the pipelined attention kernel's time goes - timing-only ablation builds (one round) and
# SQ / cache counter passes of the default build
OUT=$PWD/gpurun_out; REPO=$PWD; mkdir -p $OUT
AB_ROUNDS=1 bash tools/ab_bench.sh abl1 abl2 abl3 abl4 abl5
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_r4b_$i -o pmc -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-f16x1 --no-train --no-extra > $OUT/pmc_r4b_$i.log 2>&1
  echo "pmc $i exit $?"
  cd $REPO; python tools/rocpd_summary.py pmc $(ls $OUT/pmc_r4b_$i/*results.db $OUT/pmc_r4b_$i/*/*results.db 2>/dev/null | head -1) > $OUT/pmc_r4b_$i.txt; cd /tmp
  grep -E '^kernel|attn_sp|conv3x_kernelILi2|dwpw2s' $OUT/pmc_r4b_$i.txt | cut -c1-400
  rm -rf $OUT/pmc_r4b_$i
done
