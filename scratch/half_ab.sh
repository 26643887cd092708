#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { # label, env...
  label=$1; shift
  env "$@" python3 bench.py --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline --no-config4 --no-config3 --no-config5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$label', 'K1 alone us', round(d['kernels']['voxel_accumulate']['ms_avg']*1e3,1), 'step us', round(d['ms_per_step']*1e3,1), 'Gpts/s', round(d['value']/1e3,1), 'call+count', {k:round(v,1) for k,v in d.get('call_then_count_us',{}).items() if k!='note'}, {k:round(v['ms_avg']*1e3,1) for k,v in d['kernels'].items()})"
}
run full X=0
run half CWIPC_K1_TABLE=1024
run full X=0
run half CWIPC_K1_TABLE=1024
run half_spare12 CWIPC_K1_TABLE=1024 CWIPC_SPARE_CUS=12
run full_spare12 CWIPC_SPARE_CUS=12
