#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -p no:cacheprovider --maxfail=6 -k "conformer or attention or tscnet or batch_rows" > $OUT/pytest_gpu_r2g.log 2>&1
echo "pytest exit $?"; grep -E "passed|failed|error" $OUT/pytest_gpu_r2g.log | tail -3; grep -E "^FAILED|^E  " $OUT/pytest_gpu_r2g.log | head -20
bash tools/ab_bench.sh "$@" 2>&1 | tail -10
