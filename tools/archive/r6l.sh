#!/bin/bash
# round 6 session l: forward STFT as a real FFT (stft_fft400_kernel): parity, then time against the folded-DFT kernel
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stft or amplitude or round_trip or enhance_batch_matches or tscnet_stages or streaming_enhancer" 2>&1 | tail -4
for v in 1 0; do echo "CMGAN_STFT_FFT=$v"; CMGAN_STFT_FFT=$v timeout 300 python tools/batch_sweep.py 2>/dev/null | grep stft_compress; done
