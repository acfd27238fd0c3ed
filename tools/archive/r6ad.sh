#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_training.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python tools/train_bench.py --batches 32 --steps 5 --adversarial > $OUT/r6ad_train.json 2>$OUT/r6ad_train.err; tail -1 $OUT/r6ad_train.json | cut -c1-1500
bash tools/train_trace.sh r06 > $OUT/r6ad_trace.log 2>&1; head -30 $OUT/train_trace_r06.txt | cut -c1-110
