#!/bin/bash
# Round profile: rocprofv3 kernel trace + stats of the default bench, then two PMC passes (HBM bytes).
# usage (on the GPU box): bash scratch/profile_round.sh r01
R=${1:-r01}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/profile_$R
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/bench_trace.err || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline > $OUT/bench_write.json 2> $OUT/bench_write.err || exit 1
python3 scratch/profile_summary.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
