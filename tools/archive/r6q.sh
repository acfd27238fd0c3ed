#!/bin/bash
# round 6 session q: pipelined streaming (decoders of step k beside the encoder / TSCBs of step k + 1)
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stream_config5.py -m gpu -x -q -k "stream or config5" 2>&1 | tail -3
timeout 300 python tools/stream_bench.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k: v['ms_per_10s_clip'] for k, v in d['results'].items()})"
