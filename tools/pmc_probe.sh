#!/bin/bash
# Extra SQ counter passes for a kernel-level stall breakdown.  Usage: bash tools/pmc_probe.sh <tag> [bench args]
TAG=${1:-probe}; shift
OUT=$PWD/gpurun_out; mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $OUT/avail_$TAG.txt 2>&1
for C in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
         "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAVES"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-24)
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_${TAG}_$N -o pmc -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > $OUT/pmc_${TAG}_$N.log 2>&1
  echo "rocprof pmc $N exit $?"
done
