#!/bin/bash
# Round-end session: full GPU parity suite, smoke, the default bench line (with f32_mode + cpu_baseline), the 48 kHz
# bench line, the front-end batch sweep and the config-5 streaming bench.  Outputs are copied into profiles/ by hand.
TAG=${1:-r02}
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider --maxfail=8 > $OUT/pytest_gpu_$TAG.log 2>&1
echo "pytest exit $?"; grep -E "passed|failed|error" $OUT/pytest_gpu_$TAG.log | tail -3; grep -E "^FAILED|^E  " $OUT/pytest_gpu_$TAG.log | head -20
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -2
timeout 600 python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench exit $?"; head -c 600 $OUT/bench_$TAG.json; echo
timeout 600 python bench.py --workload 48k --no-cpu-baseline --steps 5 --warmup 2 > $OUT/bench_${TAG}_48k.json 2> $OUT/bench_${TAG}_48k.err; echo "bench48k exit $?"; head -c 400 $OUT/bench_${TAG}_48k.json; echo
timeout 300 python tools/batch_sweep.py > $OUT/batch_sweep_$TAG.txt 2>&1; tail -14 $OUT/batch_sweep_$TAG.txt
timeout 300 python tools/stream_bench.py > $OUT/stream_$TAG.json 2> $OUT/stream_$TAG.err; cat $OUT/stream_$TAG.json
