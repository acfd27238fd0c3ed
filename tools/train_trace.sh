#!/bin/bash
# rocprofv3 kernel trace of the adversarial training step at 32 clips per GPU (tools/train_bench.py): per-kernel durations
# -> gpurun_out/train_trace_<tag>.txt (copied into profiles/ in the build container)
TAG=${1:-r06}
OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_train_$TAG -o trace -- python $REPO/tools/train_bench.py --batches 32 --steps 2 --adversarial > $OUT/prof_train_$TAG.log 2>&1; echo "trace $?"
cd $REPO
DB=$(ls $OUT/prof_train_$TAG/*/*_results.db 2>/dev/null | head -1); [ -z "$DB" ] && DB=$(find $OUT/prof_train_$TAG -name "*.db" | head -1)
python tools/rocpd_summary.py trace $DB > $OUT/train_trace_$TAG.txt; head -70 $OUT/train_trace_$TAG.txt | cut -c1-120
rm -rf $OUT/prof_train_$TAG
