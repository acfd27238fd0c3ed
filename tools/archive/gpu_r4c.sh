#!/bin/bash
# round 4, session c: query tiles per block of the pipelined attention (L2 working set vs pipeline fill)
AB_ROUNDS=1 bash tools/ab_bench.sh tpb1 tpb2 tpb3 tpb4
