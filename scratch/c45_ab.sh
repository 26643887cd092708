#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { label=$1; shift
  env "$@" python3 bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-config3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); c4=d['config4']; c5=d['config5']
print('$label', 'config4 ms/frame', round(c4['ms_per_frame'],3), 'config5 fps', round(c5['value']), 'p50', round(c5['p50_ms'],3), 'p99', round(c5['p99_ms'],2), 'host', round(c5['host_arrays_in_and_out_ms_per_frame'],2), 'pinned', round(c5['page_locked_arrays_in_and_out_ms_per_frame'],2))"
}
run ws3 X=1
run ws2 CWIPC_WORKSPACES=2
run old CWIPC_LIBRARY_DIR=$GRAFT_REPO_ROOT/scratch/lib_old
run ws3 X=1
run ws2 CWIPC_WORKSPACES=2
