#!/bin/bash
# Same-box A/B of the whole forward: another build of the library (the previous commit, or a -D variant such as -DPF_DMA_B) (profiles/scratch/libs/libpriorflow_prev.so,
# built from a worktree of HEAD) against the current build, interleaved.
for v in prev cur prev cur; do
  if [ $v = prev ]; then export PRIORFLOW_LIB=$PWD/profiles/scratch/libs/libpriorflow_prev.so; else unset PRIORFLOW_LIB; fi
  python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])"
done
