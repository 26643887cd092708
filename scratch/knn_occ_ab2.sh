#!/bin/bash
cd $GRAFT_REPO_ROOT
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
for n,v in d.items():
    if int(n) > 100000: print('   n', n, 'ms', round(v['ms'],3), 'knn', v['kernels_ms'].get('sor_knn_mean_dist'))"; }
for i in 1 2; do
echo "== 6 waves, four loads in flight (tree)"; python3 scratch/sor_bench.py 2>/dev/null | show
echo "== 6 waves, two in flight"; CWIPC_LIBRARY_DIR=$GRAFT_REPO_ROOT/scratch/lib_va python3 scratch/sor_bench.py 2>/dev/null | show
echo "== 7 waves, two in flight"; CWIPC_LIBRARY_DIR=$GRAFT_REPO_ROOT/scratch/lib_vb python3 scratch/sor_bench.py 2>/dev/null | show
echo "== 7 waves, four in flight"; CWIPC_LIBRARY_DIR=$GRAFT_REPO_ROOT/scratch/lib_vc python3 scratch/sor_bench.py 2>/dev/null | show
done
