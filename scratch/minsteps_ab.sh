#!/bin/bash
cd $GRAFT_REPO_ROOT
for m in 1 8 12 16 1 8 16; do
echo "== CWIPC_K1_MIN_STEPS=$m"; CWIPC_K1_MIN_STEPS=$m python scratch/mid_size.py 2>/dev/null | cut -c1-150
done
