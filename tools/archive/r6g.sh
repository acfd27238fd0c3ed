#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT
CMGAN_HIP_LIB=$PWD/cmgan_amd/lib/variants/pcstamp/libcmgan_hip.so python tools/probes/pc_stamps.py 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tscnet_stages or config2 or enhance_batch_matches or shard or stream_encoder" 2>&1 | tail -2
bash tools/knob_sweep.sh "CMGAN_CONV_PC=0" "CMGAN_CONV_PC=1" 2>&1 | tee $OUT/r6g_conv_pc_ab.txt
