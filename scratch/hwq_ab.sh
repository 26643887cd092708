#!/bin/bash
cd $GRAFT_REPO_ROOT
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
for k,v in d.items():
    if 'no_stage' in k or k=='resident': print('  ', k, round(v['fps'],1), 'fps, p50', round(v['p50_ms'],3), 'ms')"; }
for q in default 8 16; do
  echo "== GPU_MAX_HW_QUEUES=$q"
  if [ $q = default ]; then python scratch/chain_bench.py 2>/dev/null | show; else GPU_MAX_HW_QUEUES=$q python scratch/chain_bench.py 2>/dev/null | show; fi
done
