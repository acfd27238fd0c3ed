#!/bin/bash
# round 6 session j: mask / complex decoder as two parallel paths: parity, streaming time, same-session A/B vs a -DDEC_PARALLEL=0 build
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stream_config5.py -m gpu -x -q -k "tscnet or config2 or references_own or enhance or stream or shard or graph or branched or windowed or real_recordings or config5" 2>&1 | tail -3
timeout 300 python tools/stream_bench.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k: v['ms_per_10s_clip'] for k, v in d['results'].items()})"
CMGAN_HIP_LIB=$PWD/cmgan_amd/lib/variants/decseq/libcmgan_hip.so timeout 300 python tools/stream_bench.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('decseq', {k: v['ms_per_10s_clip'] for k, v in d['results'].items()})"
AB_ROUNDS=2 bash tools/ab_bench.sh decseq 2>&1 | cut -c1-150 | tee $OUT/r6j_dec_parallel_ab.txt
