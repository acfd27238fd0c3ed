#!/bin/bash
# round 4, session d: which fetches the pipelined attention waits for (E table vs K / V images), and the L2 / fabric
# counters of the 1-tile-per-block build against the default 6
OUT=$PWD/gpurun_out; REPO=$PWD; mkdir -p $OUT
AB_ROUNDS=1 bash tools/ab_bench.sh abl6 abl7
cd /tmp && export TMPDIR=/tmp
for V in default tpb1; do
  if [ "$V" = default ]; then unset CMGAN_HIP_LIB; else export CMGAN_HIP_LIB=$REPO/cmgan_amd/lib/variants/$V/libcmgan_hip.so; fi
  timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_r4d_$V -o pmc -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-f16x1 --no-train --no-extra > $OUT/pmc_r4d_$V.log 2>&1
  echo "pmc $V exit $?"
  cd $REPO; python tools/rocpd_summary.py pmc $(ls $OUT/pmc_r4d_$V/*results.db $OUT/pmc_r4d_$V/*/*results.db 2>/dev/null | head -1) > $OUT/pmc_r4d_$V.txt; cd /tmp
  grep -E '^kernel|attn_sp' $OUT/pmc_r4d_$V.txt | cut -c1-300
  rm -rf $OUT/pmc_r4d_$V
done
