#!/bin/bash
# Round profile, everything from ONE invocation at ONE commit: two PMC passes for HBM bytes (first: the bench line quotes the
# traffic record they make), the bench line (defaults), rocprofv3 kernel trace + stats of the bench's headline workload, the
# issue-side counters of the accumulate kernel.  The files a round commits are written ready-named into
# gpurun_out/profile_<round>/<round>_*: copy them to profiles/ as they are.
# usage (GPU box): bash scratch/profile_round.sh r03
R=${1:-r03}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/profile_$R
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
# (the sub-records for configs 3, 4 and 5 launch the same kernels on other sizes: left out under the profiler, so that the
# per-kernel figures are those of the headline workload)
FLAGS="--no-cpu-baseline --no-config4 --no-config3 --no-config5"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --steps 8 --warmup 2 $FLAGS > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --steps 8 --warmup 2 $FLAGS > $OUT/bench_write.json 2> $OUT/bench_write.err || exit 1
python3 scratch/profile_summary.py $OUT $R --traffic-only || exit 1
cp $OUT/${R}_traffic.json profiles/${R}_traffic.json      # (this box's copy of the tree: the bench line below quotes it; commit the same file)
# the bench line as the driver runs it (all sub-records, CPU baseline)
python3 bench.py > $OUT/${R}_bench_line.json 2> $OUT/bench_line.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 200 --warmup 20 $FLAGS > $OUT/bench_trace.json 2> $OUT/bench_trace.err || exit 1
# r4: in the timed region up to three accumulate kernels are in flight at once (three workspaces and streams per thread): a kernel's
# own duration in the trace above is then longer than alone although calls complete faster.  The same trace with ONE workspace
# (CWIPC_WORKSPACES=1: every kernel of a call behind the call before, nothing overlaps): what the kernels take on their own.
CWIPC_WORKSPACES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_one -- python3 bench.py --steps 200 --warmup 20 $FLAGS > $OUT/bench_trace_one.json 2> $OUT/bench_trace_one.err || exit 1
cp $(find $OUT/trace_one -name "*kernel_stats.csv" | head -1) $OUT/${R}_bench_kernel_stats_one_workspace.csv
python3 scratch/profile_summary.py $OUT $R > $OUT/${R}_bench_rocprofv3_summary.txt || exit 1
bash scratch/pmc_k1_r02.sh pmc_k1_$R > $OUT/pmc_k1.log 2>&1
cp gpurun_out/pmc_k1_$R.txt $OUT/${R}_k1_pmc.txt
cat $OUT/${R}_bench_rocprofv3_summary.txt
ls $OUT/${R}_*
