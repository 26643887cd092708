#!/bin/bash
for s in 0 25 0 25; do
  echo "== STAGGER=$s"
  env CWIPC_K1_STAGGER=$s python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-config3 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('  K1 alone us %.1f step us %.1f config4 ms %.3f config5 fps %.0f' % (d['roofline']['kernel_ms_avg'] * 1e3, d['ms_per_step'] * 1e3, d['config4']['ms_per_frame'], d['config5']['value']))
"
done
