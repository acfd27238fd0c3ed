#!/bin/bash
# round 4, session g: EXACT fabric bytes per kernel from the request-size-resolved L2 counters (reads: 32 / 64 / 128 B
# requests; writes: 32 / 64 B), to settle whether ffn / ffn_post really fetch 1.77x / 1.36x their input
OUT=$PWD/gpurun_out; REPO=$PWD; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
         "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_r4g_$i -o pmc -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-f16x1 --no-train --no-extra > $OUT/pmc_r4g_$i.log 2>&1
  echo "pmc $i exit $?"
  cd $REPO; python tools/rocpd_summary.py pmc $(ls $OUT/pmc_r4g_$i/*results.db $OUT/pmc_r4g_$i/*/*results.db 2>/dev/null | head -1) > $OUT/pmc_r4g_$i.txt; cd /tmp
  cut -c1-260 $OUT/pmc_r4g_$i.txt | head -16
  rm -rf $OUT/pmc_r4g_$i
done
