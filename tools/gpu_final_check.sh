#!/bin/bash
# last check of a round after host-side changes (no kernel source touched: the committed PMC profiles stay valid): the GPU suite, smoke,
# the default bench line -> gpurun_out/bench_<tag>.json
TAG=${1:-r06}
OUT=$PWD/gpurun_out; mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider ) > $OUT/pytest_gpu_${TAG}_final.log 2>&1
grep -E "passed|failed|^real" $OUT/pytest_gpu_${TAG}_final.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time timeout 900 python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err ) 2>&1 | grep real; tail -c 400 $OUT/bench_$TAG.json
