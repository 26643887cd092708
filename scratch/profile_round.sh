#!/bin/bash
# Round profile: rocprofv3 kernel trace + stats of the bench's headline workload, two PMC passes for HBM bytes, and the
# issue-side counters of the accumulate kernel.  usage (GPU box): bash scratch/profile_round.sh r02
R=${1:-r02}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/profile_$R
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
# (the sub-records for configs 3 and 4 launch the same kernels on other sizes: left out, so that the per-kernel averages are
# those of the headline workload)
FLAGS="--no-cpu-baseline --no-config4 --no-config3"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 200 --warmup 20 $FLAGS > $OUT/bench_trace.json 2> $OUT/bench_trace.err || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --steps 8 --warmup 2 $FLAGS > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --steps 8 --warmup 2 $FLAGS > $OUT/bench_write.json 2> $OUT/bench_write.err || exit 1
python3 scratch/profile_summary.py $OUT > $OUT/summary.txt
bash scratch/pmc_k1_r02.sh pmc_k1_$R > $OUT/pmc_k1.log 2>&1
cp gpurun_out/pmc_k1_$R.txt $OUT/k1_pmc.txt
cat $OUT/summary.txt
