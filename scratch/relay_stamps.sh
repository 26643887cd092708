#!/bin/bash
# debug-knob build: per-workgroup stamps of the relay kernel for some sub-range sizes, with and without the flush's work
cd $GRAFT_REPO_ROOT
export CWIPC_LIBRARY_DIR=$GRAFT_REPO_ROOT/scratch/lib_dbg
for cfg in ${CFGS:-"80 0" "80 8" "120 0" "120 8" "40 8"}; do
  set -- $cfg
  echo "#### sub $1 dbg $2"
  CWIPC_RELAY_SUB=$1 CWIPC_FAST_DBG=$2 CWIPC_FAST_STAMPS=1 CWIPC_FAST_STAMPS_FILE=$GRAFT_REPO_ROOT/gpurun_out/wg_relay_$1_$2.txt timeout -k 10 120 python3 scratch/k1_phases.py 2>&1 | grep -i "relay:\|K1 by\|workgroups:" | tail -3
done
