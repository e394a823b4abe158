#!/bin/bash
# A/B of the lookup kernel variants inside ONE gpurun call (boxes differ by +-5 %): BASE = 3 taps per call, a pair load per
# tap row (the round's first version); TPT3 / TPT9 = row pairs reused between consecutive taps, 3 / 9 taps per call.
for v in BASE TPT3 TPT9 BASE TPT9; do
  echo "== $v"; PF_LIB=profiles/scratch/libs/libpf_elem_$v.so python profiles/microbench_lookup.py 200 | grep lookup
done
