#!/bin/bash
# training kernels: parity tests of the training suite, then the generator-step split at batch 4 and 32
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_gpu_training.py -x -q 2>&1 | tail -4
python tools/train_bench.py --batches 4,32 --steps 3 > gpurun_out/gen_split.json 2> gpurun_out/gen_split.err
echo "rc $?"; python - <<'P'
import json
d=json.load(open('gpurun_out/gen_split.json'))
for k,v in d['results'].items():
    print(k, v['ms_per_step'], v['kernel_ms'])
P
