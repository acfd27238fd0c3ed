#!/bin/bash
# fused vs three-core attention backward: parity of the attention tests, then the step split for both
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_gpu_training.py -x -q -k "attention or conformer_block or kink_free" 2>&1 | tail -3
for m in cores fused; do
python tools/train_bench.py --batches 4,32 --steps 3 --attn-bwd $m > gpurun_out/gen_split_$m.json 2> gpurun_out/gen_split_$m.err
echo "$m rc $?"; python - <<P
import json
d=json.load(open('gpurun_out/gen_split_$m.json'))
for k,v in d['results'].items():
    print('$m', k, v['ms_per_step'], {a:b for a,b in v['kernel_ms'].items() if a.startswith('attn')})
P
done
