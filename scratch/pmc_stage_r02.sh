#!/bin/bash
# VALU / SALU / LDS instructions of the fast accumulate kernel per wave step with stages switched off (debug-knob library)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_stage2
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export CWIPC_LIBRARY_DIR=$GRAFT_REPO_ROOT/scratch/lib_dbg
for a in 0 8 9 10 12; do
  export CWIPC_FAST_DBG=$a
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY --output-format csv -d $OUT/a$a -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config4 --no-config3 --no-config5 > $OUT/a$a.log 2>&1 || exit 1
done
python3 - <<'PY'
import csv, glob, collections, os
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc_stage2'
for d in sorted(glob.glob(out+'/a*/'), key=lambda p: int(os.path.basename(p.rstrip('/'))[1:])):
    for f in sorted(glob.glob(d+'/**/*counter_collection.csv', recursive=True)):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            agg[row['Kernel_Name']][row['Counter_Name']].append(float(row['Counter_Value']))
        for k,v in agg.items():
            if 'voxel_accumulate' in k:
                print(os.path.basename(d.rstrip('/')), {c: round(sum(x)/len(x)/39056, 1) for c,x in sorted(v.items())})
PY
