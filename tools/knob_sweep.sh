#!/bin/bash
# Same-session sweep of launch-shape knobs (environment overrides read by the launchers, e.g. CMGAN_ASP_*):
#   tools/knob_sweep.sh "A=1,B=2" "A=3" ...      (through gpurun; "-" = the defaults)
# Two passes over the list (second in reverse order) so that clock drift inside the session shows up as a spread.
OUT=gpurun_out; mkdir -p $OUT
run() {
  local combo="$1" tag="$2"
  ( if [ "$combo" != "-" ]; then for kv in $(echo "$combo" | tr ',' ' '); do export "$kv"; done; fi
    timeout 300 python bench.py --no-cpu-baseline --no-f32 --no-f16x1 --no-train --no-extra --steps 10 --warmup 3 > $OUT/knob_$tag.json 2>$OUT/knob_$tag.err )
  python - "$combo" "$OUT/knob_$tag.json" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    k = d["kernels_ms_per_step"]
    print(f"{sys.argv[1]:>48}  {d['ms_per_step']:.2f} ms  " + " ".join(f"{n}={v:.2f}" for n, v in list(k.items())[:8]))
except Exception as e:
    print(f"{sys.argv[1]:>48}  FAILED {e}")
PY
}
i=0
for c in "$@"; do i=$((i+1)); run "$c" "a$i"; done
i=0
for c in $(printf '%s\n' "$@" | tac); do i=$((i+1)); run "$c" "b$i"; done
