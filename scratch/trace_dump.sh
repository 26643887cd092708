#!/bin/bash
# kernel timeline of a stream of downsample calls, direct and with the tables left to the merge kernel (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for m in 0 2; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_dump$m
  rm -rf $OUT; mkdir -p $OUT
  CWIPC_K1_DUMP=$m rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-config4 --no-config3 --no-config5 > $OUT/bench.log 2>&1
  python3 - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = {'voxel_accumulate': 'K1', 'voxel_merge': 'MG', 'octree_replay': 'K2', 'rank_emit': 'EM'}
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), next((v for k, v in names.items() if k in r['Kernel_Name']), None)) for r in rows]
ev = [e for e in ev if e[2]]
k1 = [e for e in ev if e[2] == 'K1']
i0 = 120
t0 = k1[i0][0]
print(sys.argv[1], 'mean K1 start-to-start over 60 calls: %.1f us' % ((k1[i0 + 60][0] - k1[i0][0]) / 60e3))
for s, e, n in ev:
    if t0 <= s <= t0 + 400e3:
        print('  %s %7.1f .. %7.1f (%5.1f)' % (n, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
PY
done
