#!/bin/bash
# round 6 session d: producer / consumer dense conv (conv3x_pc_kernel): parity of the TSCNet path, then same-session A/B
# through CMGAN_CONV_PC; the f16mix band test
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --durations=8 \
  -k "tscnet or config2 or references_own or enhance_batch or stream_encoder or shard or f16mix or f16x1 or real_recordings or 48k_full" > $OUT/r6d_pytest.txt 2>&1
tail -15 $OUT/r6d_pytest.txt
bash tools/knob_sweep.sh "CMGAN_CONV_PC=0" "CMGAN_CONV_PC=1" "CMGAN_CONV_PC=1,CMGAN_BRANCHES=1" "CMGAN_CONV_PC=0,CMGAN_BRANCHES=1" 2>&1 | tee $OUT/r6d_conv_pc_ab.txt
