#!/bin/bash
# round 3, second evidence session (training side changed after the first; the inference kernels and their PMC evidence did
# not): default bench line, generator / adversarial step splits, training-step trace and matrix-pipe counters at batch 4
OUT=$PWD/gpurun_out; REPO=$PWD; mkdir -p $OUT
timeout 400 python bench.py > $OUT/bench_r03b.json 2> $OUT/bench_r03b.err; echo "bench $?"
timeout 300 python tools/train_bench.py --batches 4,32 --steps 3 > $OUT/train_r03b.json 2>/dev/null; echo "train_bench $?"
timeout 300 python tools/train_bench.py --adversarial --batches 4,32 --steps 3 > $OUT/train_adv_r03b.json 2>/dev/null; echo "adv $?"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_train_r03b -o trace -- python $REPO/tools/train_bench.py --batches 4 --steps 2 > $OUT/prof_train_r03b.log 2>&1; echo "train trace $?"
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace -d $OUT/pmc_train_r03b -o pmc -- python $REPO/tools/train_bench.py --batches 4 --steps 1 > $OUT/pmc_train_r03b.log 2>&1; echo "train pmc $?"
cd $REPO
python tools/rocpd_summary.py trace $(ls $OUT/prof_train_r03b/*results.db $OUT/prof_train_r03b/*/*results.db 2>/dev/null | head -1) > $OUT/train_trace_r03b.txt
python tools/rocpd_summary.py pmc $(ls $OUT/pmc_train_r03b/*results.db $OUT/pmc_train_r03b/*/*results.db 2>/dev/null | head -1) > $OUT/train_pmc_r03b.txt
rm -rf $OUT/prof_train_r03b $OUT/pmc_train_r03b
tail -c 1500 $OUT/bench_r03b.json
