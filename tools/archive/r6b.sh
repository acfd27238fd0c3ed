#!/bin/bash
# round 6 session b: new tests (branched form, config-5 named shape, streaming), branch-count sweep, stream bench
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_stream_config5.py tests/test_gpu_parity.py -m gpu -x -q --durations=15 \
  -k "branched or stream or config5 or windowed or graph or reloading" > $OUT/r6b_pytest.txt 2>&1
tail -25 $OUT/r6b_pytest.txt
bash tools/knob_sweep.sh "CMGAN_BRANCHES=1" "CMGAN_BRANCHES=2" "CMGAN_BRANCHES=3" "CMGAN_BRANCHES=4" "CMGAN_BRANCHES=8" "CMGAN_BRANCHES=2,CMGAN_BRANCH_OFFSET=2" 2>&1 | tee $OUT/r6b_branch_sweep.txt
timeout 600 python tools/stream_bench.py > $OUT/r6b_stream_bench.txt 2>&1; tail -20 $OUT/r6b_stream_bench.txt
