#!/bin/bash
# round 3 session j: split-f16 FeedForward / weight-gradient training kernels: parity, then A/B of the generator step
V=$PWD/cmgan_amd/lib/variants
timeout 1500 python -m pytest tests/test_gpu_training.py -x -q -m gpu 2>&1 | grep -v Warn | tail -6
echo "=== x3"; timeout 300 python tools/train_bench.py --batches 4 --steps 5 | python -c "import json,sys; d=json.load(sys.stdin)['results']['batch4']; print(d['ms_per_step'], d['kernel_ms'])"
echo "=== fp32"; CMGAN_HIP_LIB=$V/trainf32/libcmgan_hip.so timeout 300 python tools/train_bench.py --batches 4 --steps 5 | python -c "import json,sys; d=json.load(sys.stdin)['results']['batch4']; print(d['ms_per_step'], d['kernel_ms'])"
