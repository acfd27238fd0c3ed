#!/bin/bash
# round 3 session h: default attention with buffer-load addressing + coalesced distance planes vs the previous build
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conformer or attention or tscnet_stages or reproducible or hipgraph" 2>&1 | tail -3
AB_ROUNDS=2 bash tools/ab_bench.sh prev 2>&1 | cut -c1-110
