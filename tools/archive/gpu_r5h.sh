#!/bin/bash
# round 5, session h: FeedForward on 32x32x16 MFMAs (ffn32_x3_kernel): parity, then same-session A/B through CMGAN_FFN32
python -m pytest tests/test_gpu_parity.py -x -q -k "conformer or tscnet or f16x1_kernels or config2" 2>&1 | tail -4
bash tools/knob_sweep.sh - CMGAN_FFN32=0
