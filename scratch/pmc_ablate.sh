#!/bin/bash
# VALU/SALU/LDS instruction counts of K1 with stages switched off (CWIPC_VOXEL_ABLATE)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_abl
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for a in 0 1 2 4 8; do
  export CWIPC_VOXEL_ABLATE=$a
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/a$a -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/a$a.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, os
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc_abl'
for a in (0,1,2,4,8):
    for f in sorted(glob.glob(out+f'/a{a}/**/*counter_collection.csv', recursive=True)):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            agg[row['Kernel_Name']][row['Counter_Name']].append(float(row['Counter_Value']))
        for k,v in agg.items():
            if 'voxel_accumulate' in k:
                print('ablate', a, {c: round(sum(x)/len(x)/39056, 1) for c,x in v.items()}, 'per wave step; dispatches', len(next(iter(v.values()))))
PY
