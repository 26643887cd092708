#!/bin/bash
# kernel time of the fast accumulate kernel with stages switched off (debug-knob library; results are wrong, timing only)
cd $GRAFT_REPO_ROOT
export CWIPC_LIBRARY_DIR=$GRAFT_REPO_ROOT/scratch/lib_dbg
for d in 0 8 1 9 2 10 4 12; do
  CWIPC_FAST_DBG=$d python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-config4 --no-config3 --no-config5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('dbg', $d, 'K1 ms', round(d['kernels']['voxel_accumulate']['ms_avg']*1e3,1), 'step', round(d['ms_per_step']*1e3,1))"
done
