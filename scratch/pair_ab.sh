#!/bin/bash
# A/B of the accumulate kernel's workgroup shapes (round 4): CWIPC_K1_PAIR=0 one 16-wave workgroup per CU, 1 two 8-wave workgroups with
# sub-ranges by place on the CU, 2 sub-ranges by workgroup number, 3 two workgroups without sub-ranges
cd $GRAFT_REPO_ROOT
for p in ${PAIRS:-0 1 2 3 1 0}; do
  CWIPC_K1_PAIR=$p python3 bench.py --steps ${STEPS:-100} --warmup 20 --no-cpu-baseline --no-config4 --no-config3 --no-config5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('pair', $p, 'K1 alone us', round(d['kernels']['voxel_accumulate']['ms_avg']*1e3,1), 'step us', round(d['ms_per_step']*1e3,1), 'Gpts/s', round(d['value']/1e3,1), {k:round(v['ms_avg']*1e3,1) for k,v in d['kernels'].items()})"
done
