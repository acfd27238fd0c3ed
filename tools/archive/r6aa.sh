#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_training.py -x -q -m gpu 2>&1 | tail -6
for v in "1 1" "1 0" "1 1" "0 0"; do
  set -- $v
  CMGAN_FFN_BWD_FUSED=$1 CMGAN_WGRAD_LDS=$2 timeout 600 python tools/train_bench.py --batches 32 --steps 4 --adversarial > $OUT/r6aa_train.json 2>$OUT/r6aa_train.err
  python - "$v" $OUT/r6aa_train.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    r = d["results"]["batch32"]
    k = r["kernel_ms"]
    print(f"ffn_fused,wgrad_lds={sys.argv[1]}  {r['ms_per_step']:.2f} ms/step  " + " ".join(f"{n}={v:.2f}" for n, v in k.items() if "wgrad" in n or n.startswith("ffn_train_bwd")))
except Exception as e:
    print("FAILED", sys.argv[1], e, open(sys.argv[2].replace('.json', '.err')).read()[-600:])
PY
done
