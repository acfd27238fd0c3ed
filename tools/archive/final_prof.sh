OUT=$PWD/gpurun_out; REPO=$PWD; TAG=r01k
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o trace -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-f16x1 > $OUT/prof_$TAG.log 2>&1; echo trace $?
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-24)
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_${TAG}_$N -o pmc -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f16x1 > $OUT/pmc_${TAG}_$N.log 2>&1; echo pmc $N $?
done
