#!/bin/bash
# round 6 session m: STFT / ISTFT as real FFTs: parity, then time against the folded-DFT kernels
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stream_config5.py -m gpu -x -q -k "stft or amplitude or round_trip or enhance or tscnet_stages or streaming_enhancer or real_recordings or one_track or stream or windowed or config5 or smoke or pipeline" 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for v in 1 0; do echo "CMGAN_STFT_FFT=$v"; CMGAN_STFT_FFT=$v timeout 300 python tools/batch_sweep.py 2>/dev/null | grep -E "stft|B= 32"; done
