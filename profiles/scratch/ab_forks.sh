#!/bin/bash
# Same-box sweep of the fork mask (PRIORFLOW_FORKS: 1 encoders, 2 the three chains, 4 branch B's lookups, 8 head tails)
for v in 15 11 7 13 15 11; do
  PRIORFLOW_FORKS=$v python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('forks $v', d['value'], d['ms_per_step'])"
done
