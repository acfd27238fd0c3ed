#!/bin/bash
# round 6 session a: GPU suite with durations (which tests make up the 477 s) + two-branch graph sweep
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q --durations=60 > $OUT/r6a_pytest.txt 2>&1
tail -3 $OUT/r6a_pytest.txt
bash tools/knob_sweep.sh "-" "CMGAN_BRANCHES=2,CMGAN_BRANCH_OFFSET=0" "CMGAN_BRANCHES=2,CMGAN_BRANCH_OFFSET=4" "CMGAN_BRANCHES=2,CMGAN_BRANCH_OFFSET=12" "CMGAN_BRANCHES=2,CMGAN_BRANCH_OFFSET=24" "CMGAN_BRANCHES=2,CMGAN_BRANCH_OFFSET=36" "CMGAN_BRANCHES=2,CMGAN_BRANCH_OFFSET=60" 2>&1 | tee $OUT/r6a_branch_sweep.txt
