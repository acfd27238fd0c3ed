#!/bin/bash
# round 4, session f: the new parity tests (f16x1 bands; kink-free adversarial / 48 kHz / B = 4 x T = 321 steps) and the
# default bench line with the f16x1 leg
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "f16x1 or mask_matches" -s 2>&1 | grep -E "parity|passed|failed|Error|error" | tail -12
timeout 1500 python -m pytest tests/test_gpu_training.py -m gpu -q -p no:cacheprovider -k "twin" -s 2>&1 | grep -E "twin|passed|failed|Error|error" | tail -30
timeout 600 python bench.py --no-cpu-baseline --no-train --no-extra > $OUT/bench_r4f.json 2> $OUT/bench_r4f.err; echo "bench $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_r4f.json"))
print(d["ms_per_step"], d["value"], json.dumps(d.get("f16x1_mode")), json.dumps(d.get("f32_mode"))[:200])
PY
