#!/bin/bash
OUT=$PWD/gpurun_out; REPO=$PWD; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  CMGAN_STFT_FFT=$v timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_fft$v -o trace -- python $REPO/tools/probes/fft_time.py > $OUT/prof_fft$v.log 2>&1
  cd $REPO; python tools/rocpd_summary.py trace $(ls $OUT/prof_fft$v/*results.db $OUT/prof_fft$v/*/*results.db 2>/dev/null | head -1) | grep -i "fft\|fold\|ola\|kernel " ; cd /tmp
  rm -rf $OUT/prof_fft$v
done
