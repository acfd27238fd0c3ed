#!/bin/bash
# per-kernel times of the attention backward at batch 32: three cores vs fused, split by sequence axis (launch order)
OUT=$PWD/gpurun_out; REPO=$PWD; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for m in cores fused; do
rm -rf $OUT/prof_attn_$m
timeout 400 rocprofv3 --kernel-trace -d $OUT/prof_attn_$m -o trace -- python $REPO/tools/train_bench.py --batches 32 --steps 1 --attn-bwd $m > $OUT/prof_attn_$m.log 2>&1; echo "trace $m $?"
python - <<P
import sqlite3,glob,collections
db=glob.glob('$OUT/prof_attn_$m/**/*results.db',recursive=True)[0]
cur=sqlite3.connect(db).cursor()
tabs=[r[0] for r in cur.execute("select name from sqlite_master where type='table' or type='view'")]
kd=[t for t in tabs if 'kernel_dispatch' in t and 'rocpd' in t]
# use the 'kernels' view if present
try:
    rows=list(cur.execute("select name, start, end, grid_x, workgroup_x from kernels order by start"))
except Exception as e:
    print('no kernels view', e, tabs[:20]); rows=[]
agg=collections.defaultdict(lambda:[0,0.0])
for name,st,en,gx,wx in rows:
    if not name.startswith('at_') and 'at_' not in name[:40]: continue
    k=(name.split('(')[0][:40], gx//max(wx,1))
    agg[k][0]+=1; agg[k][1]+=(en-st)/1e3
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:14]:
    print('$m', k, v[0], round(v[1]/v[0],1),'us avg', round(v[1]/4e3,2),'ms/step')
P
rm -rf $OUT/prof_attn_$m
done
