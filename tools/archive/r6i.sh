#!/bin/bash
# round 6 session i: eager branches on CU-masked streams (cmgan_set_branch_cu_split) vs the two-branch graph
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "branched" 2>&1 | tail -2
bash tools/knob_sweep.sh "-" "CMGAN_CU_SPLIT=1" "CMGAN_CU_SPLIT=2" "CMGAN_CU_SPLIT=1,CMGAN_BRANCH_OFFSET=3" "CMGAN_CU_SPLIT=1,CMGAN_BRANCH_OFFSET=10" \
  "CMGAN_CU_SPLIT=1,CMGAN_BRANCH_OFFSET=24" "CMGAN_CU_SPLIT=2,CMGAN_BRANCH_OFFSET=10" "CMGAN_CU_SPLIT=1,CMGAN_BRANCHES=4,CMGAN_BRANCH_OFFSET=5" 2>&1 | cut -c1-120 | tee $OUT/r6i_cu_split.txt
