#!/bin/bash
# round 3 session b: multi-tile 32x32x16 attention: parity, cycle stamps (TPB 6 and 1), A/B vs tpb1 / attn16, one PMC pass
OUT=gpurun_out; mkdir -p $OUT
V=$PWD/cmgan_amd/lib/variants
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conformer or attention or tscnet_stages or reproducible or hipgraph" 2>&1 | tail -5
echo "=== stamps TPB=6"; CMGAN_HIP_LIB=$V/a32stamp/libcmgan_hip.so timeout 200 python tools/probes/attn_stamps.py 2>&1 | tail -24
echo "=== stamps TPB=1"; CMGAN_HIP_LIB=$V/a32stamp1/libcmgan_hip.so timeout 200 python tools/probes/attn_stamps.py 2>&1 | tail -24
AB_ROUNDS=2 bash tools/ab_bench.sh tpb1 attn16 2>&1 | tail -8
REPO=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $REPO/$OUT/pmc_r3b -o pmc -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-f16x1 --no-train > $REPO/$OUT/pmc_r3b.log 2>&1; echo "pmc exit $?"
cd $REPO; python tools/rocpd_summary.py pmc $OUT/pmc_r3b/pmc_results.db 2>&1 | head -12 | cut -c1-260 || ls -R $OUT/pmc_r3b | head
