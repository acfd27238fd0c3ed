#!/bin/bash
# round 6 session e: producer / consumer dense conv after the counted-wait fix: parity subset, A/B through CMGAN_CONV_PC
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tscnet_stages or config2 or enhance_batch_matches or shard or 48k_variant or stream_encoder" > $OUT/r6e_pytest.txt 2>&1
tail -4 $OUT/r6e_pytest.txt
bash tools/knob_sweep.sh "CMGAN_CONV_PC=0" "CMGAN_CONV_PC=1" 2>&1 | tee $OUT/r6e_conv_pc_ab.txt
