#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2; do
echo "== before (no scout)"; CWIPC_LIBRARY_DIR=$GRAFT_REPO_ROOT/scratch/lib_head python scratch/mid_size.py 2>/dev/null
echo "== scout"; python scratch/mid_size.py 2>/dev/null
done
run() { label=$1; shift
  env "$@" python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-config4 --no-config3 --no-config5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$label', 'K1 alone us', round(d['kernels']['voxel_accumulate']['ms_avg']*1e3,1), 'step us', round(d['ms_per_step']*1e3,2), 'call+count', {k:round(v,1) for k,v in d.get('call_then_count_us',{}).items() if k!='note'})"; }
run before CWIPC_LIBRARY_DIR=$GRAFT_REPO_ROOT/scratch/lib_head
run scout X=1
run before CWIPC_LIBRARY_DIR=$GRAFT_REPO_ROOT/scratch/lib_head
run scout X=1
