#!/bin/bash
# fused FeedForward backward (part A + both weight gradients on the chip): parity, then same-session A/B of the training step
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_training.py -x -q -m gpu -k "feed_forward or fused_residual or three_adamw or whole_conformer or tscb_trains or generator_train_step_matches or graphed_train_step" 2>&1 | tail -8
for v in 1 0 1 0; do
  CMGAN_FFN_BWD_FUSED=$v timeout 600 python tools/train_bench.py --batches 32 --steps 4 --adversarial > $OUT/r6y_train_$v.json 2>$OUT/r6y_train_$v.err
  python - $v $OUT/r6y_train_$v.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    r = d["results"]["batch32"]
    k = r["kernel_ms"]
    print(f"fused={sys.argv[1]}  {r['ms_per_step']:.2f} ms/step  peak {r.get('peak_mem_GB')} GB  " + " ".join(f"{n}={v:.2f}" for n, v in k.items() if n.startswith("ffn")))
except Exception as e:
    print("FAILED", sys.argv[1], e, open(sys.argv[2].replace('.json', '.err')).read()[-600:])
PY
done
