#!/bin/bash
# one bench line per configuration (no CPU baseline), condensed to a table
for c in C1 C2 C3 C4 C5 C2nd C4t1 C1j C1t; do
  python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-5s ms_per_solve=%8.3f  problem-iterations/s=%12.0f  nonfinite=%.3f  %s' % ('$c', d['ms_per_step'], d['value'], d['nonfinite_frac'], d['config']['workload']))"
done
