#!/bin/bash
O=gpurun_out/r04j; mkdir -p $O; export TMPDIR=/tmp
rm -rf $O/prof; timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o p -f csv -- python bench_ops.py --only knn_tree --iters 3 --out $O/knn_tree_prof.json > /dev/null 2>&1
python - <<'P'
import csv
rows=list(csv.DictReader(open('gpurun_out/r04j/prof/p_kernel_trace.csv')))
import collections
agg=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name']
    if 'knn_tree' in n:
        agg[(n.split('(')[0][:60], r['Grid_Size_X'], r['Grid_Size_Y'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in agg.items(): print(k, 'n',len(v),'median us', sorted(v)[len(v)//2])
P
rm -rf $O/prof
