#!/bin/bash
# round 5, session b: dense conv tile / weight-prefetch variants (same-session A/B) + parity of the candidates
AB_ROUNDS=2 bash tools/ab_bench.sh c128 c128w2 cw2
for v in c128w2; do
  CMGAN_HIP_LIB=$PWD/cmgan_amd/lib/variants/$v/libcmgan_hip.so python -m pytest tests/test_gpu_parity.py -x -q -k "tscnet or config2 or enhance" 2>&1 | tail -3
done
python -m pytest tests/test_gpu_training.py -x -q -k "trainer" 2>&1 | tail -3
