#!/bin/bash
# Profiling session: rocprofv3 kernel trace + separate PMC passes of the default bench command; summaries go to
# gpurun_out/ and are turned into profiles/<tag>_* by tools/rocpd_summary.py in the build container.
TAG=${1:-r03}
OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
timeout 300 python bench.py --no-cpu-baseline --no-f32 --no-f16x1 > $OUT/bench_${TAG}_prof.json 2> $OUT/bench_${TAG}_prof.err; echo "bench $?"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o trace -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-f32 --no-f16x1 --no-train --no-extra > $OUT/prof_$TAG.log 2>&1; echo "trace $?"
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-24)
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_${TAG}_$N -o pmc -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-f16x1 --no-train --no-extra > $OUT/pmc_${TAG}_$N.log 2>&1; echo "pmc $N $?"
done
cd $REPO; du -sh $OUT/prof_$TAG $OUT/pmc_${TAG}_* 2>/dev/null
