#!/bin/bash
# VALU/SALU/LDS instructions of K1 per wave step with stages switched off one after another
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_stage
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for a in ${ABLS:-0 2 4 20 52 116 8}; do
  export CWIPC_VOXEL_ABLATE=$a
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_BRANCH --output-format csv -d $OUT/a$a -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/a$a.log 2>&1 || exit 1
done
python3 - <<'PY'
import csv, glob, collections, os
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc_stage'
for d in sorted(glob.glob(out+'/a*/'), key=lambda p: int(os.path.basename(p.rstrip('/'))[1:])):
    for f in sorted(glob.glob(d+'/**/*counter_collection.csv', recursive=True)):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            agg[row['Kernel_Name']][row['Counter_Name']].append(float(row['Counter_Value']))
        for k,v in agg.items():
            if 'voxel_accumulate' in k:
                print(os.path.basename(d.rstrip('/')), {c: round(sum(x)/len(x)/39056, 1) for c,x in v.items()})
PY
