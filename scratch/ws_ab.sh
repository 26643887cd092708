#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { # label, env...
  label=$1; shift
  env "$@" python3 bench.py --steps ${STEPS:-200} --warmup 30 --no-cpu-baseline --no-config4 --no-config3 --no-config5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$label', 'K1 alone us', round(d['kernels']['voxel_accumulate']['ms_avg']*1e3,1), 'step us', round(d['ms_per_step']*1e3,1), 'Gpts/s', round(d['value']/1e3,1), 'call+count', {k:round(v,1) for k,v in d.get('call_then_count_us',{}).items() if k!='note'})"
}
run ws2 CWIPC_WORKSPACES=2
run ws3 CWIPC_WORKSPACES=3
run ws4 CWIPC_WORKSPACES=4
run ws3_dump2 CWIPC_WORKSPACES=3 CWIPC_K1_DUMP=2
run ws4_dump2 CWIPC_WORKSPACES=4 CWIPC_K1_DUMP=2
run ws4_dump2_spare16 CWIPC_WORKSPACES=4 CWIPC_K1_DUMP=2 CWIPC_SPARE_CUS=16
run ws3_spare16 CWIPC_WORKSPACES=3 CWIPC_SPARE_CUS=16
run ws3_spare0 CWIPC_WORKSPACES=3 CWIPC_SPARE_CUS=0
run ws2 CWIPC_WORKSPACES=2
run ws3 CWIPC_WORKSPACES=3
