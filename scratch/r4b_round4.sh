#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== chain (config 5)"; python scratch/chain_bench.py 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
for k,v in d.items(): print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a!='stage_ms_per_frame'}, {a:round(b,3) for a,b in v.get('stage_ms_per_frame',{}).items()})"
echo "== stamps (debug-knob build)"
export CWIPC_LIBRARY_DIR=$GRAFT_REPO_ROOT/scratch/lib_dbg CWIPC_FAST_STAMPS=1
CWIPC_FAST_STAMPS_FILE=$GRAFT_REPO_ROOT/gpurun_out/stamps_small.txt python3 scratch/k1_phases.py 2> gpurun_out/stamps_small.log
grep -a "====\|K1 by events\|debug:" gpurun_out/stamps_small.log | head -40
