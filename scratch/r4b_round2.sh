#!/bin/bash
cd $GRAFT_REPO_ROOT
for cpp in 16 32 64; do for t in 0.3 0.5; do
  echo "== SPARSE_CELLS_PER_POINT=$cpp CELL_TARGET=$t"; CWIPC_SOR_SPARSE_CELLS_PER_POINT=$cpp CWIPC_SOR_CELL_TARGET=$t python3 scratch/sor_bench.py 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
for n,v in d.items():
    if int(n) > 100000: print('  n', n, 'ms', round(v['ms'],3), 'knn', v['kernels_ms'].get('sor_knn_mean_dist'), 'sum', round(sum(v['kernels_ms'].values()),3))"
done; done
echo "== chain (config 5)"; python scratch/chain_bench.py 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
for k,v in d.items(): print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a!='stage_ms_per_frame'}, {a:round(b,3) for a,b in v.get('stage_ms_per_frame',{}).items()})"
