#!/bin/bash
# round 5, session a: new parity tests + attention launch-shape sweep
python -m pytest tests/test_gpu_parity.py -x -q -k "references_own_modules or f16x1_kernels_on_edge or oversized" 2>&1 | tail -5
python -m pytest tests/test_gpu_training.py -x -q -k "mask_stream_offsets or trainer" 2>&1 | tail -5
bash tools/knob_sweep.sh - CMGAN_ASP_GROUP_SHORT=1 CMGAN_ASP_GROUP_SHORT=1,CMGAN_ASP_ALIGN_SHORT=1 CMGAN_ASP_GROUP_SHORT=4 CMGAN_ASP_GROUP_SHORT=16 CMGAN_ASP_GROUP_SHORT=1,CMGAN_ASP_ALIGN_SHORT=1,CMGAN_ASP_SLOTS=1024 CMGAN_ASP_TPB_LONG=3 CMGAN_ASP_TPB_LONG=6,CMGAN_ASP_GROUP_LONG=32
