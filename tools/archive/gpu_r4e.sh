#!/bin/bash
# round 4, session e: A/B after the per-slot refills (+ 4 tiles per block), per-axis launch times of the pipelined and the
# 16x16x32 attention kernels from a kernel trace
OUT=$PWD/gpurun_out; REPO=$PWD; mkdir -p $OUT
AB_ROUNDS=2 bash tools/ab_bench.sh attn16
cd /tmp && export TMPDIR=/tmp
for V in default attn16; do
  if [ "$V" = default ]; then unset CMGAN_HIP_LIB; else export CMGAN_HIP_LIB=$REPO/cmgan_amd/lib/variants/$V/libcmgan_hip.so; fi
  timeout 300 rocprofv3 --kernel-trace -d $OUT/tr_r4e_$V -o tr -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32 --no-f16x1 --no-train --no-extra > $OUT/tr_r4e_$V.log 2>&1
  cd $REPO; python - $(ls $OUT/tr_r4e_$V/*results.db $OUT/tr_r4e_$V/*/*results.db 2>/dev/null | head -1) <<'PY'
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = db.execute(f"select s.kernel_name, d.end - d.start, d.grid_size_x from {kd} d join {ks} s on d.kernel_id = s.id").fetchall()
agg = collections.defaultdict(list)
for n, dur, g in rows:
    if 'attn' in n: agg[(n[:40], g)].append(dur / 1000.0)
for k, v in sorted(agg.items()): print(k, len(v), 'avg us %.1f' % (sum(v[len(v)//2:]) / len(v[len(v)//2:])))
PY
  cd /tmp; rm -rf $OUT/tr_r4e_$V
done
