#!/bin/bash
# Round 4, second session: the small clouds' flow of the outlier filter (ten launches, one-workgroup scan, shells beyond the first with a bound per row)
# against the twelve-launch flow, and the cell size under both.
cd $GRAFT_REPO_ROOT
line() { grep "points" | sed 's/\(.* points: [0-9.]* us per call\).*sor_knn_mean_dist.: \([0-9.]*\).*sum \([0-9.]*\)/   \1, knn \2 us, kernels \3/'; }
for sm in 524288 0; do for t in 0.3 0.4 0.5 0.65; do
  echo "== SMALL_CELLS=$sm CELL_TARGET=$t"; env CWIPC_SOR_SMALL_CELLS=$sm CWIPC_SOR_CELL_TARGET=$t python scratch/sor_small.py 2>&1 | line
done; done
echo "== default, all kernels"; python scratch/sor_small.py 2>&1 | grep points
