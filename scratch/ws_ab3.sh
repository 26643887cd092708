#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { # label, env...
  label=$1; shift
  env "$@" python3 bench.py --steps ${STEPS:-200} --warmup 30 --no-cpu-baseline --no-config4 --no-config3 --no-config5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$label', 'K1 alone us', round(d['kernels']['voxel_accumulate']['ms_avg']*1e3,1), 'step us', round(d['ms_per_step']*1e3,1), 'Gpts/s', round(d['value']/1e3,1), 'call+count', {k:round(v,1) for k,v in d.get('call_then_count_us',{}).items() if k!='note'})"
}
run ws3 X=1
run full8_ws3 CWIPC_K1_PAIR=2
run full8_ws4 CWIPC_K1_PAIR=2 CWIPC_WORKSPACES=4
run full8_ws3_spare0 CWIPC_K1_PAIR=2 CWIPC_SPARE_CUS=0
run full8_ws4_spare0 CWIPC_K1_PAIR=2 CWIPC_WORKSPACES=4 CWIPC_SPARE_CUS=0
run full8_ws2 CWIPC_K1_PAIR=2 CWIPC_WORKSPACES=2
run ws3 X=1
