#!/bin/bash
# round 5, session f: STFT bin-block split sweep (kernel durations by rocprofv3, one session)
OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
python -m pytest tests/test_gpu_parity.py -x -q -k "stft or istft or rms" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
for S in 3 7 13; do
  CMGAN_STFT_BSPLIT=$S timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_stft_$S -o trace -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32 --no-f16x1 --no-train --no-extra > $OUT/prof_stft_$S.log 2>&1
  echo "== BSPLIT $S (rc $?)"
  python $REPO/tools/rocpd_summary.py trace $(find $OUT/prof_stft_$S -name "*.db" | head -1) | grep -E "stft|irfft|ola|rms"
done
