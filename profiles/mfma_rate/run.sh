#!/bin/bash
# hipcc --offload-arch=gfx950 -O3 profiles/mfma_rate/mfma_rate.hip -o profiles/mfma_rate/mfma_rate   (build container), then on the GPU box:
./profiles/mfma_rate/mfma_rate | tee gpurun_out/r3_mfma_rate.txt
