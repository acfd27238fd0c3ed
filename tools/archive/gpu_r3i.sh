#!/bin/bash
# round 3 session i: the tightened parity tests (kink-free twin whole-step, real-track floors, 48 kHz both modes, graphed step)
timeout 1500 python -m pytest tests -x -q -m gpu -k "kink_free or real_recordings or 48k_full_size or graphed or adversarial_train_step or generator_train_step" 2>&1 | grep -v Warning | tail -30
