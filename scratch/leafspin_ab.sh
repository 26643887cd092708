#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2; do
echo "== old (HEAD)"; CWIPC_LIBRARY_DIR=$GRAFT_REPO_ROOT/scratch/lib_old python scratch/mid_size.py 2>/dev/null
echo "== new (id looked at alone)"; python scratch/mid_size.py 2>/dev/null
done
