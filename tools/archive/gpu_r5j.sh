#!/bin/bash
# round 5, session j: dense conv on 32x32x16 MFMAs (conv3x32_kernel, CMGAN_CONV32=1): parity, then same-session A/B
CMGAN_CONV32=1 python -m pytest tests/test_gpu_parity.py -x -q -k "tscnet_stages or tscnet_full or config2 or references_own or f16x1_mode" 2>&1 | tail -4
bash tools/knob_sweep.sh - CMGAN_CONV32=1
