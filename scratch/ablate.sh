#!/bin/bash
# K1 ablation study: time voxel_accumulate with parts switched off (results are wrong then; timing only)
for a in 0 1 2 18 22 6 4 8 12 16; do
  CWIPC_VOXEL_ABLATE=$a timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernels']
        print('ablate=$a', 'K1 %.1f us' % (k['voxel_accumulate']['ms_avg']*1000), ' wall %.1f us' % (d['ms_per_step']*1000), ' outputs', d['config']['outputs_per_gpu'])
"
done
