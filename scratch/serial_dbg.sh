#!/bin/bash
# kernel time of the serial accumulate kernel with stages switched off (debug-knob library; results are wrong, timing only)
# knobs (CWIPC_SERIAL_DBG): 8192 loads + boxes only; 4096 + LDS staging; 2048|1024 + the lane walk without boundaries; 1024 + boundaries and merge, entries dropped; 16384 no merge; 8 no flush
cd $GRAFT_REPO_ROOT
export CWIPC_LIBRARY_DIR=$GRAFT_REPO_ROOT/scratch/lib_dbg
export CWIPC_VOXEL_SERIAL=1
for d in 0 8 16384 16392 1024 1032 3072 3080 4096 4104 8192 8200; do
  CWIPC_SERIAL_DBG=$d python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-config4 --no-config3 --no-config5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('dbg', $d, 'K1 us', round(d['kernels']['voxel_accumulate']['ms_avg']*1e3,1), 'step', round(d['ms_per_step']*1e3,1))"
done
CWIPC_FAST_STAMPS=1 python3 scratch/k1_phases.py 2>&1 | grep -i "debug\|K1 by\|====" | tail -12
