#!/bin/bash
python -m pytest tests/test_gpu_dist_rccl.py -x -q 2>&1 | tail -15
