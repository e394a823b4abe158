#!/bin/bash
# A/B of the iteration-boundary schedule (same box, interleaved): default forks | strips serial on the calling stream (FORKS=7) |
# both lookups as one launch (LOOKUP_PAIR=1) | both
for i in 1 2; do
for cfg in "PRIORFLOW_FORKS=15 PRIORFLOW_LOOKUP_PAIR=0" "PRIORFLOW_FORKS=7 PRIORFLOW_LOOKUP_PAIR=0" "PRIORFLOW_FORKS=15 PRIORFLOW_LOOKUP_PAIR=1" "PRIORFLOW_FORKS=7 PRIORFLOW_LOOKUP_PAIR=1" "PRIORFLOW_FORKS=3 PRIORFLOW_LOOKUP_PAIR=1"; do
  env $cfg python bench.py --no-cpu-baseline --steps 60 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', d['value'], 'pairs/s', d['ms_per_step'], 'ms')"
done; done
