#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { # label, env...
  label=$1; shift
  env "$@" python3 bench.py --steps ${STEPS:-200} --warmup 30 --no-cpu-baseline --no-config4 --no-config3 --no-config5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$label', 'K1 alone us', round(d['kernels']['voxel_accumulate']['ms_avg']*1e3,1), 'step us', round(d['ms_per_step']*1e3,1), 'Gpts/s', round(d['value']/1e3,1))"
}
run ws3 X=1
run ws3_pair CWIPC_K1_PAIR=1
run ws3_pair_spare0 CWIPC_K1_PAIR=1 CWIPC_SPARE_CUS=0
run ws3_pair_spare16 CWIPC_K1_PAIR=1 CWIPC_SPARE_CUS=16
run ws4_pair CWIPC_K1_PAIR=1 CWIPC_WORKSPACES=4
run ws3_spare4 CWIPC_SPARE_CUS=4
run ws3_spare12 CWIPC_SPARE_CUS=12
run ws3_spare24 CWIPC_SPARE_CUS=24
run ws3 X=1
