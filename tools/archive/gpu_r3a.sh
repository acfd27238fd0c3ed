#!/bin/bash
# round 3 session a: parity of the 32x32x16 attention + A/B against the 16x16x32 kernels (variant attn16)
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conformer or attention or tscnet or reproducible or hipgraph or real_recordings or native" 2>&1 | tail -15
AB_ROUNDS=2 bash tools/ab_bench.sh attn16 2>&1 | tail -20
