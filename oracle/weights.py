"""Seeded weights / clips shared by the oracle tests and the product benchmark.

TEST INFRASTRUCTURE (see oracle/__init__.py): a re-export of ``cmgan_amd.synth`` so that the
tests, the golden generator and the GPU path all draw the same bytes from the same PCG64 streams.
"""
from cmgan_amd.synth import (C, CONV_EXP, CONV_K, DIM_HEAD, FF_MULT, HEADS, MAX_POS,  # noqa: F401
                             conformer_state_dict, make_state_dict, synthetic_clips)
