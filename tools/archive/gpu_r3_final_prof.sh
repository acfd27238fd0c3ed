#!/bin/bash
# round 3: the evidence session - default bench line, kernel trace + PMC passes of the same command (tools/gpu_prof.sh),
# training-step trace and matrix-pipe counters, batch sweep
bash tools/gpu_prof.sh r03
OUT=$PWD/gpurun_out; REPO=$PWD
timeout 300 python tools/train_bench.py --batches 4,32 --steps 3 > $OUT/train_r03.json 2>/dev/null; echo "train_bench $?"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_train_r03 -o trace -- python $REPO/tools/train_bench.py --batches 4 --steps 2 > $OUT/prof_train_r03.log 2>&1; echo "train trace $?"
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace -d $OUT/pmc_train_r03 -o pmc -- python $REPO/tools/train_bench.py --batches 4 --steps 1 > $OUT/pmc_train_r03.log 2>&1; echo "train pmc $?"
cd $REPO
timeout 300 python tools/batch_sweep.py > $OUT/batch_sweep_r03.txt 2>/dev/null; echo "sweep $?"
