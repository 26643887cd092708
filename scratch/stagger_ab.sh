#!/bin/bash
# the accumulate kernel with ranges of growing length (CWIPC_K1_STAGGER = per cent; _REV: the longest ranges to the first workgroups):
# alone (hipEvents), a call in the stream, call-then-count
for cfg in "0 0" "20 0" "25 0" "30 0" "20 1" "25 1" "30 1" "0 0" "25 0" "25 1"; do
  set -- $cfg
  echo "== STAGGER=$1 REV=$2"
  env CWIPC_K1_STAGGER=$1 CWIPC_K1_STAGGER_REV=$2 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-config4 --no-config3 --no-config5 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('  K1 alone us %.1f step us %.1f Gpts/s %.1f call+count %.1f parity %s' % (d['roofline']['kernel_ms_avg'] * 1e3, d['ms_per_step'] * 1e3, d['value'] / 1e3, d.get('call_then_count_us', {}).get('+0.01'), d.get('parity', {}).get('rgb_tile_exact')))
"
done
