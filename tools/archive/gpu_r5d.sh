#!/bin/bash
python -m pytest tests/test_gpu_parity.py -x -q -k "stream" 2>&1 | tail -4
timeout 600 python bench.py --no-cpu-baseline --no-f32 --no-f16x1 --no-train --steps 10 --warmup 3 > gpurun_out/bench_r5d.json 2> gpurun_out/bench_r5d.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_r5d.json"))
print(d["ms_per_step"], json.dumps(d.get("stream_config5"), indent=1))
PY
tail -5 gpurun_out/bench_r5d.err
