#!/bin/bash
cd $GRAFT_REPO_ROOT
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
for n,v in d.items(): print('   n', n, 'ms', round(v['ms'],3), 'knn', v['kernels_ms'].get('sor_knn_mean_dist'))"; }
for i in 1 2; do
echo "== HEAD (four loads, then the four candidates)"; CWIPC_LIBRARY_DIR=$GRAFT_REPO_ROOT/scratch/lib_head python3 scratch/sor_bench.py 2>/dev/null | show
echo "== pipelined, 5 waves per SIMD (spills)"; python3 scratch/sor_bench.py 2>/dev/null | show
echo "== pipelined, 4 waves per SIMD"; CWIPC_LIBRARY_DIR=$GRAFT_REPO_ROOT/scratch/lib_old python3 scratch/sor_bench.py 2>/dev/null | show
done
echo "== 2 M"; 
echo "HEAD"; CWIPC_LIBRARY_DIR=$GRAFT_REPO_ROOT/scratch/lib_head python scratch/sor_small.py 2>&1 | grep "^1999396" | cut -c1-60
echo "pipelined 5"; python scratch/sor_small.py 2>&1 | grep "^1999396" | cut -c1-60
echo "pipelined 4"; CWIPC_LIBRARY_DIR=$GRAFT_REPO_ROOT/scratch/lib_old python scratch/sor_small.py 2>&1 | grep "^1999396" | cut -c1-60
