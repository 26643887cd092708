#!/bin/bash
cd $GRAFT_REPO_ROOT
line() { grep "^36262" | sed 's/\(.* points: [0-9.]* us per call\).*sor_knn_mean_dist.: \([0-9.]*\).*/   \1, knn \2 us/'; }
echo "== default (pair kernel, statistics folded, coalesced scan), all kernels"; python scratch/sor_small.py 2>&1 | grep "^36262"
echo "== CWIPC_SOR_STATS_FOLD=0"; CWIPC_SOR_STATS_FOLD=0 python scratch/sor_small.py 2>&1 | grep "^36262"
for t in 0.3 0.35 0.4 0.5 0.65; do
  echo "== pair, CELL_TARGET=$t"; CWIPC_SOR_CELL_TARGET=$t python scratch/sor_small.py 2>&1 | line
done
echo "== default again"; python scratch/sor_small.py 2>&1 | grep "^36262"
