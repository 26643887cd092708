#!/bin/bash
# k-NN grid cell occupancy sweep (CWIPC_SOR_CELL_TARGET = points per occupied cell / (k + 1)); timing only
cd $GRAFT_REPO_ROOT
for t in 0.35 0.5 0.75 1.0 1.5 2.0; do
  echo "target $t"
  CWIPC_SOR_CELL_TARGET=$t python3 scratch/sor_bench.py 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
for n,v in d.items(): print('  n', n, 'ms', round(v['ms'],3), 'knn', v['kernels_ms'].get('sor_knn_mean_dist'), 'cells', v['kernels_ms'].get('sor_exclusive_scan'))"
done
