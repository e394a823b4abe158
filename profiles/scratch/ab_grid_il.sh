#!/bin/bash
# Same-box A/B: lookups with the planar grid (two 8-byte loads per row pair) vs the interleaved copy (one 16-byte load)
for v in 0 1 0 1; do
  PRIORFLOW_GRID_IL=$v python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('grid_il $v', d['value'], d['ms_per_step'])"
done
