#!/bin/bash
# kernel trace of the generator training step at batch 32 (2 warm-up + 1 timed + 1 profiled-by-label step = 4 steps)
OUT=$PWD/gpurun_out; REPO=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_train_b32 -o trace -- python $REPO/tools/train_bench.py --batches 32 --steps 1 > $OUT/prof_train_b32.log 2>&1; echo "trace $?"
cd $REPO
python tools/rocpd_summary.py trace $(ls $OUT/prof_train_b32/*/*results.db $OUT/prof_train_b32/*results.db 2>/dev/null | head -1) > $OUT/train_b32_trace.txt; head -75 $OUT/train_b32_trace.txt
