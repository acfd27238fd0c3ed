#!/bin/bash
# round 4, session h: interleaved tile order of the pipelined attention (A32_GROUP = 64) against the consecutive order
# (variant grp1): parity, same-session A/B, and the exact fabric read requests of both
OUT=$PWD/gpurun_out; REPO=$PWD; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "conformer or attn or tscnet or golden or mask" 2>&1 | tail -4
AB_ROUNDS=2 bash tools/ab_bench.sh grp1
cd /tmp && export TMPDIR=/tmp
for v in default grp1; do
  if [ "$v" = default ]; then unset CMGAN_HIP_LIB; else export CMGAN_HIP_LIB=$REPO/cmgan_amd/lib/variants/$v/libcmgan_hip.so; fi
  timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/pmc_r4h_$v -o pmc -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-f16x1 --no-train --no-extra > $OUT/pmc_r4h_$v.log 2>&1
  echo "pmc $v exit $?"
  cd $REPO; python tools/rocpd_summary.py pmc $(ls $OUT/pmc_r4h_$v/*results.db $OUT/pmc_r4h_$v/*/*results.db 2>/dev/null | head -1) > $OUT/pmc_r4h_$v.txt; cd /tmp
  grep -i "attn\|kernel" $OUT/pmc_r4h_$v.txt | cut -c1-260 | head -8
  rm -rf $OUT/pmc_r4h_$v
done
