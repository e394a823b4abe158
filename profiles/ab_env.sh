#!/bin/bash
# Same-box A/B of environment configurations of the B=1 forward, interleaved.
# usage: ab_env.sh REPS "CFG_A" "CFG_B" ...   (a CFG is a string of VAR=value pairs; "-" = the defaults)
# BENCH_ARGS overrides the bench flags (default: the bare forward leg, 60 timed steps).
REPS=$1; shift
ARGS=${BENCH_ARGS:---no-cpu-baseline --no-batch32 --no-train-step --steps 60}
for i in $(seq $REPS); do
for cfg in "$@"; do
  e=$cfg; [ "$cfg" = "-" ] && e="PF_AB_DEFAULTS=1"
  env $e python bench.py $ARGS 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-40s' % '$cfg', d['value'], 'pairs/s', d['ms_per_step'], 'ms')"
done; done
