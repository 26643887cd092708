#!/bin/bash
# HBM bytes of the compaction kernels at 10 M points (tile filter, crop): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_compact
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 scratch/compact_profile.py > $OUT/fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 scratch/compact_profile.py > $OUT/write.log 2>&1 || exit 1
python3 - $OUT <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
print("# compaction kernels at 10 M points (scratch/compact_profile.py: tilefilter(1) keeps 4 999 122, crop keeps 1 121 476, masked(3) keeps all),")
print("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE x 2 for 16-byte-per-lane streaming loads on gfx950 (MI355X_MICROARCH.md); KiB per dispatch")
for name in ("fetch", "write"):
    for f in glob.glob(out + f"/{name}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "compact" in row["Kernel_Name"]:
                agg[row["Kernel_Name"].split("(")[0][-60:]].append(float(row["Counter_Value"]))
        for k, v in agg.items():
            print("%-62s %s n=%3d  values (KiB): %s" % (k, name.upper() + "_SIZE", len(v), sorted(set(round(x) for x in v))[:8]))
PY
tail -4 $OUT/fetch.log
