#!/bin/bash
# Same-session A/B of library builds: tools/ab_bench.sh <variant> [<variant> ...]   (through gpurun)
# Alternates the default build and each cmgan_amd/lib/variants/<name> build (four rounds, order reversed
# every other round) and prints
# ms/step plus the per-kernel table, so box-to-box clock differences cancel out.
OUT=gpurun_out; mkdir -p $OUT
for round in $(seq 1 ${AB_ROUNDS:-4}); do
  # order alternates between rounds: the first process of a round tends to run ~1 % slower than the last
  if [ $((round % 2)) = 1 ]; then order="default $*"; else order="$(echo default "$@" | tr ' ' '\n' | tac | tr '\n' ' ')"; fi
  for v in $order; do
    if [ "$v" = default ]; then unset CMGAN_HIP_LIB; else export CMGAN_HIP_LIB=$PWD/cmgan_amd/lib/variants/$v/libcmgan_hip.so; fi
    timeout 300 python bench.py --no-cpu-baseline --no-f32 --no-f16x1 --no-train --no-extra --steps 10 --warmup 3 > $OUT/ab_${v}_$round.json 2>/dev/null
    python - "$v" "$round" "$OUT/ab_${v}_$round.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[3]))
k = d["kernels_ms_per_step"]
print(f"{sys.argv[1]:>10} r{sys.argv[2]}  {d['ms_per_step']:.2f} ms  " + " ".join(f"{n}={v:.2f}" for n, v in list(k.items())[:11]))
PY
  done
done
