#!/bin/bash
cd $GRAFT_REPO_ROOT
for st in 1 0; do for t in 0.5 0.75 1.0 1.25 1.5 2.0 3.0; do
  echo "== STAGED=$st CELL_TARGET=$t"; env CWIPC_SOR_STAGED=$st CWIPC_SOR_CELL_TARGET=$t python scratch/sor_small.py 2>&1 | grep "points" | sed 's/\(.* points: [0-9.]* us per call\).*sor_knn_mean_dist.: \([0-9.]*\).*/   \1, knn \2 us/'
done; done
