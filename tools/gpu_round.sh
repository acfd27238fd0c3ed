#!/bin/bash
# One GPU-box session: parity suite, default bench line, rocprofv3 kernel trace + PMC passes.
# Usage (from the repo root, through gpurun): bash tools_gpu_round.sh <tag>
TAG=${1:-r01}
OUT=$PWD/gpurun_out
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $OUT/pytest_gpu_$TAG.log 2>&1
echo "pytest exit $?"; grep -E "passed|failed" $OUT/pytest_gpu_$TAG.log | tail -2
timeout 600 python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
echo "bench exit $?"; cat $OUT/bench_$TAG.json
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o trace -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_$TAG.log 2>&1
echo "rocprof trace exit $?"
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-24)
  timeout 600 rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_${TAG}_$N -o pmc -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc_${TAG}_$N.log 2>&1
  echo "rocprof pmc $N exit $?"
done
cd $REPO
find $OUT -name "*.csv" | head -40
du -sh $OUT
