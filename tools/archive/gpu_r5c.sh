#!/bin/bash
python -m pytest tests/test_gpu_parity.py -x -q -k "stream" 2>&1 | tail -30
