#!/bin/bash
# round 3 session g: attn32 epilogue ablations (timing only) and tiles-per-block sweep
AB_ROUNDS=1 bash tools/ab_bench.sh noepi nobar tpb1 tpb3 tpb12 2>&1 | cut -c1-60
