#!/bin/bash
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/k2twice
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export CWIPC_K2_TWICE=1
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err || exit 1
python3 - <<'PY'
import csv, glob, os
f=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/k2twice/trace/**/*kernel_trace.csv', recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'octree_replay' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
d=[int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in rows]
first=d[0::2][-20:]; second=d[1::2][-20:]
print('first of pair', sum(first)/len(first), 'second of pair', sum(second)/len(second))
PY
