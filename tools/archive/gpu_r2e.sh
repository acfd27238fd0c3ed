#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_parity.py -m gpu -q -s -p no:cacheprovider --maxfail=6 -k "validation or data_path or tscnet or batch_rows or real_rec or hipgraph" > $OUT/pytest_gpu_r2e.log 2>&1
echo "pytest exit $?"; grep -E "passed|failed|error" $OUT/pytest_gpu_r2e.log | tail -3; grep -E "^FAILED|^E  " $OUT/pytest_gpu_r2e.log | head -20
grep "parity\]" $OUT/pytest_gpu_r2e.log | grep -i "validation" | head
bash tools/ab_bench.sh "$@" 2>&1 | tail -10
