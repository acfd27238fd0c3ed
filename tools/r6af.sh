#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_training.py -x -q -m gpu 2>&1 | tail -2
timeout 600 python tools/train_bench.py --batches 32 --steps 4 --adversarial > $OUT/r6af_train.json 2>$OUT/r6af_train.err
python - $OUT/r6af_train.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["results"]["batch32"]
print(f"{r['ms_per_step']:.2f} ms/step  " + " ".join(f"{n}={v:.2f}" for n, v in r["kernel_ms"].items()))
PY
