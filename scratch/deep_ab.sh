#!/bin/bash
# the accumulate kernel with two steps in flight behind the one being processed (-DCWIPC_K1_DEEP, scratch/lib_deep) against the shipped one
for lib in "" scratch/lib_deep "" scratch/lib_deep; do
  echo "== lib=${lib:-shipped}"
  env ${lib:+CWIPC_LIBRARY_DIR=$PWD/$lib} python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-config4 --no-config3 --no-config5 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('  K1 alone us %.1f step us %.1f Gpts/s %.1f call+count %.1f parity %s' % (d['roofline']['kernel_ms_avg'] * 1e3, d['ms_per_step'] * 1e3, d['value'] / 1e3, d.get('call_then_count_us', {}).get('+0.01'), d.get('parity', {}).get('rgb_tile_exact')))
"
done
