#!/bin/bash
# Round-2 session B: full GPU parity suite, default bench line, fused-attention A/B.
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -s -p no:cacheprovider --maxfail=6 > $OUT/pytest_gpu_r2b.log 2>&1
echo "pytest exit $?"; grep -E "passed|failed|error" $OUT/pytest_gpu_r2b.log | tail -3; grep -E "^FAILED|^E  " $OUT/pytest_gpu_r2b.log | head -20
timeout 600 python bench.py > $OUT/bench_r2b.json 2> $OUT/bench_r2b.err
echo "bench exit $?"; python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/bench_r2b.json") if l.startswith("{")][-1])
print(d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d.get("f32_mode",{}).get("ms_per_step"))
print(d["kernels_ms_per_step"])
PY
bash tools/ab_bench.sh "$@" 2>&1 | tail -12
