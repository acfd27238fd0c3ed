#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_training.py -x -q -m gpu 2>&1 | tail -2
for v in 1 2; do
  timeout 600 python tools/train_bench.py --batches 32 --steps 4 --adversarial > $OUT/r6bl_train.json 2>$OUT/r6bl_train.err
  python - $v $OUT/r6bl_train.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r = d["results"]["batch32"]
print(f"run {sys.argv[1]}  {r['ms_per_step']:.2f} ms/step")
PY
done
