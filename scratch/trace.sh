#!/bin/bash
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace1
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench.log 2>&1
find $OUT -name "*.csv" | head
S=$(find $OUT -name "*kernel_stats.csv" | head -1)
echo "== $S"; cat $S | cut -c1-200 | head -20
