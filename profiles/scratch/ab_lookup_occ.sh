#!/bin/bash
# Lookup kernel occupancy (amdgpu_waves_per_eu): W5 = the allocator's own choice (90 VGPRs), W6, W8 forced
for v in ${VARIANTS:-W5 W6 W8 W5 W6 W8}; do
  echo "== $v"; PF_LIB=profiles/scratch/libs/libpf_elem_$v.so python profiles/microbench_lookup.py 200 | grep lookup
done
