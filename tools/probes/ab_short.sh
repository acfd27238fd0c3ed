OUT=gpurun_out; mkdir -p $OUT
V=$PWD/cmgan_amd/lib/variants/ffnpf8/libcmgan_hip.so
CMGAN_HIP_LIB=$V timeout 120 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "conformer or tscnet" 2>&1 | tail -2
for round in 1 2; do
  for v in default ffnpf8 ; do
    if [ "$v" = default ]; then unset CMGAN_HIP_LIB; else export CMGAN_HIP_LIB=$V; fi
    timeout 120 python bench.py --no-cpu-baseline --no-f32 --no-train --steps 10 --warmup 3 > $OUT/ab_${v}_$round.json 2>/dev/null
    python - "$v" "$round" "$OUT/ab_${v}_$round.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[3]))
k = d["kernels_ms_per_step"]
print(f"{sys.argv[1]:>10} r{sys.argv[2]}  {d['ms_per_step']:.2f} ms  " + " ".join(f"{n}={v:.2f}" for n, v in list(k.items())[:9]))
PY
  done
done
