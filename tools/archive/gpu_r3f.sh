#!/bin/bash
# round 3 session f: attn32 with coalesced distance planes, to_out image in LDS, early Q / residual prefetch
OUT=gpurun_out; mkdir -p $OUT
V=$PWD/cmgan_amd/lib/variants
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conformer or attention or tscnet_stages or reproducible or hipgraph" 2>&1 | tail -5
echo "=== stamps"; CMGAN_HIP_LIB=$V/a32stamp/libcmgan_hip.so timeout 200 python tools/probes/attn_stamps.py 2>&1 | tail -32
AB_ROUNDS=2 bash tools/ab_bench.sh attn16 nold onlyld 2>&1 | cut -c1-100
