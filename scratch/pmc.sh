#!/bin/bash
# PMC passes over the bench (separate passes: SQ has 8 slots)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/a -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM --output-format csv -d $OUT/b -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/b.log 2>&1
find $OUT -name "*counter_collection.csv" | head
python3 - <<'PY'
import csv, glob, collections, os
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc1'
for f in sorted(glob.glob(out+'/*/**/*counter_collection.csv', recursive=True)):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        k=row['Kernel_Name'][:40]
        agg[k][row['Counter_Name']].append(float(row['Counter_Value']))
    print(f)
    for k,v in agg.items():
        if 'voxel_accumulate' in k or 'octree' in k:
            print(' ', k, {c: round(sum(x)/len(x)) for c,x in v.items()}, 'dispatches', len(next(iter(v.values()))))
PY
