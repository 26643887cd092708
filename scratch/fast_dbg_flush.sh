#!/bin/bash
cd $GRAFT_REPO_ROOT
export CWIPC_LIBRARY_DIR=$GRAFT_REPO_ROOT/scratch/lib_dbg
for d in ${DBG_LIST:-0 64 512 128 384 256 8}; do
  CWIPC_FAST_DBG=$d python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-config4 --no-config3 --no-config5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('dbg', $d, 'K1 us', round(d['kernels']['voxel_accumulate']['ms_avg']*1e3,1), 'step', round(d['ms_per_step']*1e3,1), 'outputs', d['config']['outputs_per_gpu'])"
done
