#!/bin/bash
# k-NN grid sweeps (timing only): CWIPC_SOR_SPARSE_CELLS_PER_POINT = virtual fine-grid cells per point at most, CWIPC_SOR_CELL_TARGET = points per occupied cell / (k + 1)
cd $GRAFT_REPO_ROOT
for cpp in 12 16 32; do
 for t in 0.3 0.4 0.5 0.65; do
  echo "cells per point $cpp target $t"
  CWIPC_SOR_SPARSE_CELLS_PER_POINT=$cpp CWIPC_SOR_CELL_TARGET=$t python3 scratch/sor_bench.py 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
for n,v in d.items():
    if int(n) > 1000000: print('  n', n, 'ms', round(v['ms'],3), 'knn', v['kernels_ms'].get('sor_knn_mean_dist'), 'mark', v['kernels_ms'].get('sor_seg_mark'))"
 done
done
