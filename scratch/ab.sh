#!/bin/bash
# A/B: bench.py with scratch/lib_old (previous build) and with the in-tree library, alternating, on the same box
for i in 1 2 3; do
  CWIPC_LIBRARY_DIR=$PWD/scratch/lib_old timeout -k 10 120 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('old', round(d['ms_per_step']*1000,2), {k:round(v['ms_avg']*1000,2) for k,v in d['kernels'].items()})" || exit 1
  timeout -k 10 120 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new', round(d['ms_per_step']*1000,2), {k:round(v['ms_avg']*1000,2) for k,v in d['kernels'].items()})" || exit 1
done
