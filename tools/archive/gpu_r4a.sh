#!/bin/bash
# round 4, session a: the software-pipelined attention - parity of everything that runs through a conformer, then
# same-session A/B against the 16x16x32 kernel (variant attn16) and the un-pipelined 32x32x16 kernel (attn32)
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5
AB_ROUNDS=2 bash tools/ab_bench.sh attn16 attn32
