#!/bin/bash
# Issue-side PMC counters of the voxel accumulate kernel per dispatch and per 256-point wave step (BASELINE configs[1]).
# usage (GPU box): bash scratch/pmc_k1_r02.sh <outname>     -> gpurun_out/<outname>.txt
NAME=${1:-pmc_k1}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$NAME
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
ARGS="bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-config4 --no-config3 --no-config5"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/p1 -- python3 $ARGS > $OUT/p1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU --output-format csv -d $OUT/p2 -- python3 $ARGS > $OUT/p2.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAVES SQ_ACTIVE_INST_MISC SQ_IFETCH --output-format csv -d $OUT/p3 -- python3 $ARGS > $OUT/p3.log 2>&1 || true
python3 - <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/$NAME.txt
import csv, glob, collections, os
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/'+os.environ.get('NAME','pmc_k1') if False else None
PY
NAME=$NAME python3 - <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/$NAME.txt
import csv, glob, collections, os
name=os.environ['NAME']
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/'+name
STEPS=39056   # 9 998 244 points / 256 per wave step
print("# rocprofv3 --pmc, python3 bench.py --steps 4 --warmup 2 (BASELINE configs[1]); mean per dispatch and per 256-point wave step (39056 steps per dispatch)")
for d in sorted(glob.glob(out+'/p*/')):
    for f in sorted(glob.glob(d+'/**/*counter_collection.csv', recursive=True)):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            agg[row['Kernel_Name']][row['Counter_Name']].append(float(row['Counter_Value']))
        for k,v in agg.items():
            if 'voxel_accumulate' in k:
                short = k.split('(')[0][-60:]
                for c,x in sorted(v.items()):
                    m=sum(x)/len(x)
                    print(f"{short:60s} {c:24s} dispatches {len(x):3d}  per dispatch {m:14.0f}  per wave step {m/STEPS:9.2f}")
PY
cat $GRAFT_REPO_ROOT/gpurun_out/$NAME.txt
