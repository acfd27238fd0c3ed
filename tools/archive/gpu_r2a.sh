#!/bin/bash
# Round-2 session A: full GPU parity suite, default bench line, XCD-order A/B, 48 kHz full-size bench.
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -s -p no:cacheprovider -x > $OUT/pytest_gpu_r2a.log 2>&1
echo "pytest exit $?"; grep -E "passed|failed|error" $OUT/pytest_gpu_r2a.log | tail -3
timeout 600 python bench.py > $OUT/bench_r2a.json 2> $OUT/bench_r2a.err
echo "bench exit $?"; python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/bench_r2a.json") if l.startswith("{")][-1])
print(d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d.get("f32_mode",{}).get("ms_per_step"), d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline",{}).get("b4"))
print(d["kernels_ms_per_step"])
PY
bash tools/ab_bench.sh noslide noxcd 2>&1 | tail -10
timeout 600 python bench.py --workload 48k --no-cpu-baseline --steps 5 --warmup 2 > $OUT/bench_r2a_48k.json 2> $OUT/bench_r2a_48k.err
echo "bench48k exit $?"; tail -c 1500 $OUT/bench_r2a_48k.json; tail -3 $OUT/bench_r2a_48k.err
