#!/bin/bash
# Timing-only ablations of pf_dccl_lookup (results are WRONG in every variant but BASE): which of the three gather
# stages (own window, grid taps, other-volume taps) or the stores bounds the kernel.  Build here, run on the GPU box:
#   cd prior-flow_amd/csrc && for v in BASE NO_OWN NO_GRID NO_OTH NO_STORE; do hipcc --offload-arch=gfx950 -O3 -ffp-contract=off \
#       -fno-slp-vectorize -fPIC -shared -DPF_ABL_$v pf_elem_kernels.hip -o ../../profiles/scratch/libs/libpf_elem_$v.so; done
for v in BASE NO_OWN NO_GRID NO_OTH NO_STORE BASE; do
  echo "== $v"; PF_LIB=profiles/scratch/libs/libpf_elem_$v.so python profiles/microbench_lookup.py 200 | grep lookup
done
