#!/usr/bin/env python3
"""Cycle stamps of conv3x_pc_kernel (a -DPC_STAMP build: cmgan_amd.build.build(variant="pcstamp", extra_flags=["-DPC_STAMP"]),
CMGAN_HIP_LIB=.../variants/pcstamp/libcmgan_hip.so): per role (producer wave 4, consumer wave 0 of every block) the cycles in
each phase, per stage.  Through gpurun."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cmgan_amd import TSCNet, _lib
from cmgan_amd.synth import make_state_dict, synthetic_clips

m = TSCNet(64, 201).load_state_dict(make_state_dict(0)).eval()
wav = synthetic_clips(32, 32000, seed=0).cuda()
lib = _lib.load()
fn = lib.cmgan_dbg_pc_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 16)()
m.engine.enhance(wav); torch.cuda.synchronize()
fn(buf, 1)
m.engine.enhance(wav); torch.cuda.synchronize()
fn(buf, 0)
v = list(buf)
gp, gc = max(v[12], 1), max(v[14], 1)
print(f"producer (per stage, cycles): write {v[0] / gp:.0f}  fetch+advance {v[1] / gp:.0f}  barrier wait {v[2] / gp:.0f}   [{gp} stages]")
print(f"consumer (per stage, cycles): barrier wait {v[4] / gc:.0f}  MFMA phase {v[5] / gc:.0f}  epilogue {v[6] / gc:.0f}   [{gc} stages]")
