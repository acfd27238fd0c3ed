#!/bin/bash
# round 3 session c: finer cycle stamps of attn32 with two waves per SIMD and with one (A32_LONE)
V=$PWD/cmgan_amd/lib/variants
echo "=== stamps 2 waves/SIMD"; CMGAN_HIP_LIB=$V/a32stamp/libcmgan_hip.so timeout 200 python tools/probes/attn_stamps.py 2>&1 | tail -32
echo "=== stamps 1 wave/SIMD"; CMGAN_HIP_LIB=$V/a32lone/libcmgan_hip.so timeout 200 python tools/probes/attn_stamps.py 2>&1 | tail -32
