#!/bin/bash
# round 3 session e: compile-time ablation of attn32 (timing only): what does each phase cost?
AB_ROUNDS=1 bash tools/ab_bench.sh nor nosm nopv nold norld onlyld 2>&1 | cut -c1-60
