#!/bin/bash
# Issue-side PMC counters of K1 (per wave step), full kernel and with the table inserts switched off
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_k1
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for a in 0 4; do
  export CWIPC_VOXEL_ABLATE=$a
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/p1a$a -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/p1a$a.log 2>&1 || exit 1
  rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU --output-format csv -d $OUT/p2a$a -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/p2a$a.log 2>&1 || exit 1
done
python3 - <<'PY'
import csv, glob, collections, os
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc_k1'
for d in sorted(glob.glob(out+'/p*a*/')):
    for f in sorted(glob.glob(d+'/**/*counter_collection.csv', recursive=True)):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            agg[row['Kernel_Name']][row['Counter_Name']].append(float(row['Counter_Value']))
        for k,v in agg.items():
            if 'voxel_accumulate' in k:
                print(os.path.basename(d.rstrip('/')), {c: round(sum(x)/len(x)/39056, 1) for c,x in v.items()}, 'per wave step')
PY
