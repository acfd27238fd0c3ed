#!/bin/bash
# discriminator kernels: parity tests, then the adversarial-step split
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_gpu_training.py -x -q -k "discriminator or adversarial or graphed or trainer" 2>&1 | tail -5
python tools/train_bench.py --adversarial --batches 4,32 --steps 3 > gpurun_out/adv_split.json 2> gpurun_out/adv_split.err
echo "rc $?"; python - <<'P'
import json
d=json.load(open('gpurun_out/adv_split.json'))
for k,v in d['results'].items():
    print(k, v['ms_per_step'], {a:b for a,b in v['kernel_ms'].items() if a.startswith('disc')})
P
