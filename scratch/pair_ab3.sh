#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { # label, env...
  label=$1; shift
  env "$@" python3 bench.py --steps ${STEPS:-100} --warmup 20 --no-cpu-baseline --no-config4 --no-config3 --no-config5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$label', 'K1 alone us', round(d['kernels']['voxel_accumulate']['ms_avg']*1e3,1), 'step us', round(d['ms_per_step']*1e3,1), 'Gpts/s', round(d['value']/1e3,1), {k:round(v['ms_avg']*1e3,1) for k,v in d['kernels'].items()})"
}
run HEAD CWIPC_LIBRARY_DIR=$GRAFT_REPO_ROOT/scratch/lib_old
run dump1 CWIPC_K1_DUMP=1
run dump0 CWIPC_K1_DUMP=0
run dump1_pair CWIPC_K1_DUMP=1 CWIPC_K1_PAIR=1
run dump1_spare16 CWIPC_K1_DUMP=1 CWIPC_SPARE_CUS=16
run dump1_spare0 CWIPC_K1_DUMP=1 CWIPC_SPARE_CUS=0
run HEAD CWIPC_LIBRARY_DIR=$GRAFT_REPO_ROOT/scratch/lib_old
run dump1 CWIPC_K1_DUMP=1
