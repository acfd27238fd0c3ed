#!/bin/bash
# Round-2 session C: training-slice tests + STFT tests, bench, front-end batch sweep, streaming bench.
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_parity.py -m gpu -q -s -p no:cacheprovider --maxfail=6 -k "train or loss or stft or istft or pipeline or round_trip or enhance_batch or smoke or silent" > $OUT/pytest_gpu_r2c.log 2>&1
echo "pytest exit $?"; grep -E "passed|failed|error" $OUT/pytest_gpu_r2c.log | tail -3; grep -E "^FAILED|^E  " $OUT/pytest_gpu_r2c.log | head -20
grep "parity\]" $OUT/pytest_gpu_r2c.log | grep -i "ffn\|stft" | head -40
timeout 300 python bench.py --no-cpu-baseline --no-f32 > $OUT/bench_r2c.json 2> $OUT/bench_r2c.err
echo "bench exit $?"; python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/bench_r2c.json") if l.startswith("{")][-1])
print(d["ms_per_step"], d["value"]); print(d["kernels_ms_per_step"]); print(d.get("stft_hbm"))
PY
timeout 300 python tools/batch_sweep.py > $OUT/batch_sweep_r2c.txt 2>&1; tail -8 $OUT/batch_sweep_r2c.txt
timeout 300 python tools/stream_bench.py > $OUT/stream_r2c.json 2> $OUT/stream_r2c.err; cat $OUT/stream_r2c.json; tail -2 $OUT/stream_r2c.err
