#!/bin/bash
cd $GRAFT_REPO_ROOT
for p in 1 0 1 0; do
  echo "== CWIPC_SOR_PAIR=$p"; CWIPC_SOR_PAIR=$p python scratch/sor_small.py 2>&1 | grep "^36262"
done
