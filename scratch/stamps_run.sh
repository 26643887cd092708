#!/bin/bash
# per-workgroup time stamps of the fast accumulate kernel (debug-knob build), 10 M points
cd $GRAFT_REPO_ROOT
export CWIPC_LIBRARY_DIR=$GRAFT_REPO_ROOT/scratch/lib_dbg CWIPC_FAST_STAMPS=1
for d in ${DUMPS:-1 0}; do
  CWIPC_K1_DUMP=$d CWIPC_FAST_STAMPS_FILE=$GRAFT_REPO_ROOT/gpurun_out/stamps_dump$d.txt python3 scratch/k1_phases.py 2> gpurun_out/stamps_dump$d.log
  grep -a "K1 by events\|debug:" gpurun_out/stamps_dump$d.log | tail -8
done
