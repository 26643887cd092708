#!/bin/bash
cd $GRAFT_REPO_ROOT
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
for n,v in d.items():
    if int(n) > 100000: print('   n', n, 'ms', round(v['ms'],3), 'knn', v['kernels_ms'].get('sor_knn_mean_dist'))"; }
for i in 1 2; do
echo "== 5 waves per SIMD (tree)"; python3 scratch/sor_bench.py 2>/dev/null | show
echo "== 6 waves per SIMD (8 registers spilled)"; CWIPC_LIBRARY_DIR=$GRAFT_REPO_ROOT/scratch/lib_w6 python3 scratch/sor_bench.py 2>/dev/null | show
echo "== 8 waves per SIMD (44 spilled)"; CWIPC_LIBRARY_DIR=$GRAFT_REPO_ROOT/scratch/lib_w8 python3 scratch/sor_bench.py 2>/dev/null | show
done
