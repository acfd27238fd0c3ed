#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
python tests/probes/mode_error.py > $OUT/mode_error_default.json 2>/dev/null; cat $OUT/mode_error_default.json
CMGAN_HIP_LIB=$PWD/cmgan_amd/lib/variants/x1/libcmgan_hip.so python tests/probes/mode_error.py > $OUT/mode_error_x1.json 2>/dev/null; cat $OUT/mode_error_x1.json
bash tools/ab_bench.sh x1 seg8 seg2 2>&1 | tail -16
