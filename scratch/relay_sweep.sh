#!/bin/bash
# kernel time of the relay accumulate kernel for several sub-range sizes (shipped library), then the debug-knob build's
# per-workgroup stamps and flush statistics
cd $GRAFT_REPO_ROOT
for sub in ${SUBS:-0 24 32 40 56 80 120}; do
  CWIPC_RELAY_SUB=$sub timeout -k 10 90 python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-config4 --no-config3 --no-config5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('sub', $sub, 'K1 us', round(d['kernels']['voxel_accumulate']['ms_avg']*1e3,1), 'step', round(d['ms_per_step']*1e3,1), 'outputs', d['config']['outputs_per_gpu'])" || exit 1
done
CWIPC_VOXEL_RELAY=0 timeout -k 10 90 python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-config4 --no-config3 --no-config5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fast variant: K1 us', round(d['kernels']['voxel_accumulate']['ms_avg']*1e3,1), 'step', round(d['ms_per_step']*1e3,1))"
export CWIPC_LIBRARY_DIR=$GRAFT_REPO_ROOT/scratch/lib_dbg
CWIPC_FAST_STAMPS=1 CWIPC_FAST_STAMPS_FILE=$GRAFT_REPO_ROOT/gpurun_out/wg_relay.txt timeout -k 10 120 python3 scratch/k1_phases.py 2>&1 | grep -i "relay\|K1 by\|====\|workgroups:" | tail -8
