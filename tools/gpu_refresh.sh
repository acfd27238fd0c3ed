#!/bin/bash
# quick refresh after a late change of an inference source: parity of the TSCNet path, bench line, kernel trace + the three
# PMC passes behind profiles/<tag>_x3_{kernel_trace_stats,pmc,hbm_traffic}
TAG=${1:-r05}
OUT=$PWD/gpurun_out; REPO=$PWD; mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py -x -q -k "tscnet or config2 or references_own or enhance or stream_encoder or shard" 2>&1 | tail -3
timeout 900 python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench exit $?"
BENCH="$REPO/bench.py --no-cpu-baseline --no-f32 --no-f16x1 --no-train --no-extra"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o trace -- python $BENCH --steps 3 --warmup 1 > $OUT/prof_$TAG.log 2>&1; echo "trace $?"
i=0
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_${TAG}_$i -o pmc -- python $BENCH --steps 1 --warmup 1 > $OUT/pmc_${TAG}_$i.log 2>&1; echo "pmc $i $?"
done
cd $REPO
db() { ls $1/*results.db $1/*/*results.db 2>/dev/null | head -1; }
python tools/rocpd_summary.py trace $(db $OUT/prof_$TAG) > $OUT/${TAG}_x3_kernel_trace_stats.txt
python tools/rocpd_summary.py pmc $(db $OUT/pmc_${TAG}_1) $(db $OUT/pmc_${TAG}_2) $(db $OUT/pmc_${TAG}_3) > $OUT/${TAG}_x3_pmc.txt
python tools/rocpd_summary.py traffic $(db $OUT/pmc_${TAG}_1) $(db $OUT/pmc_${TAG}_2) $OUT/bench_$TAG.json > $OUT/${TAG}_x3_hbm_traffic.json
rm -rf $OUT/prof_$TAG $OUT/pmc_${TAG}_*
tail -c 300 $OUT/bench_$TAG.json
