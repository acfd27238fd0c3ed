#!/bin/bash
# round 5, session e: compile-time tail (TAILK) attention variants: parity, then same-session A/B through the env knob
python -m pytest tests/test_gpu_parity.py -x -q -k "conformer or tscnet or rereference or f16x1_kernels or config2" 2>&1 | tail -4
bash tools/knob_sweep.sh - CMGAN_ASP_TAILK=0
