#!/bin/bash
# round 3 session d: is attn32 bound by operand fetches?  timing-only builds whose E / K / V loads always hit the same lines
AB_ROUNDS=2 bash tools/ab_bench.sh fakee fakekv fakeall 2>&1 | cut -c1-120
