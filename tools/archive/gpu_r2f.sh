#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_training.py -m gpu -q -p no:cacheprovider -k "data_path" 2>&1 | tail -2
for B in 2 4 8 32; do
  timeout 200 python bench.py --batch $B --no-cpu-baseline --no-f32 --steps 10 --warmup 3 > $OUT/bench_b$B.json 2>/dev/null
  python - $B <<'PY'
import json, sys
B=int(sys.argv[1])
d=json.loads([l for l in open(f"gpurun_out/bench_b{B}.json") if l.startswith("{")][-1])
k=d["kernels_ms_per_step"]
print(f"B={B:3d} {d['ms_per_step']:.2f} ms/step {d['ms_per_step']/B:.3f} ms/clip | per clip (us): " + " ".join(f"{n}={1e3*v/B:.0f}" for n,v in list(k.items())[:9]))
PY
done
