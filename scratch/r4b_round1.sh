#!/bin/bash
# second session of round 4: scan rewrite, sparse cell target, colorize without a wait
cd $GRAFT_REPO_ROOT
echo "== small clouds, default"; python scratch/sor_small.py 2>&1 | grep points
for t in 0.2 0.25 0.3 0.4 0.5; do
  echo "== CELL_TARGET=$t (10 M: sparse layout)"; CWIPC_SOR_CELL_TARGET=$t python3 scratch/sor_bench.py 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
for n,v in d.items(): print('  n', n, 'ms', round(v['ms'],3), 'knn', v['kernels_ms'].get('sor_knn_mean_dist'), 'sum', round(sum(v['kernels_ms'].values()),3))"
done
echo "== chain (config 5)"; python scratch/chain_bench.py 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
for k,v in d.items(): print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a!='stage_ms_per_frame'}, {a:round(b,3) for a,b in v.get('stage_ms_per_frame',{}).items()})"
