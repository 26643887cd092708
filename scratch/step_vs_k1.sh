#!/bin/bash
# Is the stream of calls bound by the accumulate kernel?  The kernel with stages switched off (debug-knob library; results are wrong, timing only):
# per variant the kernel alone (events) and a call in the stream (wall).
cd $GRAFT_REPO_ROOT
export CWIPC_LIBRARY_DIR=$GRAFT_REPO_ROOT/scratch/lib_dbg
for d in 0 8 12; do
  CWIPC_FAST_DBG=$d python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-config4 --no-config3 --no-config5 2> gpurun_out/step_vs_k1_$d.err | python3 -c "
import json,sys
t=sys.stdin.read()
try:
    d=json.loads(t); print('dbg', $d, 'K1 alone us', round(d['kernels']['voxel_accumulate']['ms_avg']*1e3,1), 'step us', round(d['ms_per_step']*1e3,1), {k:round(v['ms_avg']*1e3,1) for k,v in d['kernels'].items()})
except Exception as e:
    print('dbg', $d, 'failed', repr(t[:200]))"
  tail -3 gpurun_out/step_vs_k1_$d.err
done
