#!/bin/bash
# adversarial step kernel split (discriminator labels) at batch 4 and 32
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python tools/train_bench.py --adversarial --batches 4,32 --steps 3 > gpurun_out/adv_split.json 2> gpurun_out/adv_split.err
echo "rc $?"; tail -c 3000 gpurun_out/adv_split.json
