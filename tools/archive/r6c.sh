#!/bin/bash
# round 6 session c: precision ablation (F16MIX families), the full GPU suite's wall time, the default bench line
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python tools/mix_ablation.py > $OUT/r6c_mix_ablation.txt 2>&1; tail -22 $OUT/r6c_mix_ablation.txt | cut -c1-400
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 ) > $OUT/r6c_pytest.txt 2>&1
tail -22 $OUT/r6c_pytest.txt
timeout 900 python bench.py > $OUT/r6c_bench.json 2> $OUT/r6c_bench.err; tail -c 6000 $OUT/r6c_bench.json
