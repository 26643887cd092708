#!/bin/bash
# PMC counters of the outlier filter's k-NN kernel at 10 M points (sparse layout)
NAME=${1:-pmc_sor}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$NAME
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/p1 -- python3 scratch/sor_one.py > $OUT/p1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_IFETCH --output-format csv -d $OUT/p2 -- python3 scratch/sor_one.py > $OUT/p2.log 2>&1 || true
rocprofv3 --pmc TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $OUT/p3 -- python3 scratch/sor_one.py > $OUT/p3.log 2>&1 || true
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/p4 -- python3 scratch/sor_one.py > $OUT/p4.log 2>&1 || true
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 scratch/sor_one.py > $OUT/trace.log 2>&1 || true
NAME=$NAME python3 - <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/$NAME.txt
import csv, glob, collections, os
name=os.environ['NAME']
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/'+name
print("# rocprofv3 --pmc, scratch/sor_one.py (10 M points, cwipc_remove_outliers(16, 1.0)): mean per dispatch of the k-NN kernel")
for d in sorted(glob.glob(out+'/p*/')):
    for f in sorted(glob.glob(d+'/**/*counter_collection.csv', recursive=True)):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            agg[row['Kernel_Name']][row['Counter_Name']].append(float(row['Counter_Value']))
        for k,v in agg.items():
            if 'knn' in k:
                short = k.split('(')[0][-50:]
                for c,x in sorted(v.items()):
                    m=sum(x)/len(x)
                    print(f"{short:50s} {c:32s} dispatches {len(x):3d}  per dispatch {m:16.0f}")
for f in glob.glob(out+'/trace/**/*kernel_stats.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'sor' in row['Name'].lower() or 'knn' in row['Name'] or 'seg_' in row['Name'] or 'cell_' in row['Name'] or 'compact' in row['Name']:
            print(row['Name'][:70], row['Calls'], row['AverageNs'] if 'AverageNs' in row else row.get('AvgNs'))
PY
cat $GRAFT_REPO_ROOT/gpurun_out/$NAME.txt
