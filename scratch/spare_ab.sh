#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { # label, env...
  label=$1; shift
  env "$@" python3 bench.py --steps ${STEPS:-300} --warmup 30 --no-cpu-baseline --no-config4 --no-config3 --no-config5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$label', 'K1 alone us', round(d['kernels']['voxel_accumulate']['ms_avg']*1e3,1), 'step us', round(d['ms_per_step']*1e3,2), 'Gpts/s', round(d['value']/1e3,1), 'call+count', {k:round(v,1) for k,v in d.get('call_then_count_us',{}).items() if k!='note'})"
}
for i in 1 2 3; do
run spare8 CWIPC_SPARE_CUS=8
run spare12 CWIPC_SPARE_CUS=12
run spare16 CWIPC_SPARE_CUS=16
run spare24 CWIPC_SPARE_CUS=24
done
