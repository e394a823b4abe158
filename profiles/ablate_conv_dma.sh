#!/bin/bash
# Timing-only builds of pf_conv_dma_kernel (results are WRONG by construction; only the launch time is read):
#   hipcc ... -DPF_DMA_ABL_NO_DMA | -DPF_DMA_ABL_NO_READS | -DPF_DMA_ABL_NO_MFMA | -DPF_DMA_ABL_NO_BARRIER | all of NO_DMA NO_READS NO_BARRIER (= MFMAs only)
#   -c pf_conv_dma.hip, linked with the shipped objects into prior-flow_amd/lib/diag/ABL_<name>.so
# Same process environment, one launch shape after the other: the shipped kernel, then each variant (dma rows only matter).
for lib in libpriorflow_hip.so diag/ABL_NO_DMA.so diag/ABL_NO_READS.so diag/ABL_NO_MFMA.so diag/ABL_ONLY_MFMA.so diag/ABL_NO_BARRIER.so diag/ABL_NO_DMA_NO_BARRIER.so; do
  [ -f prior-flow_amd/lib/$lib ] || continue
  echo "== $lib"
  for w in zr fh1; do PRIORFLOW_LIB=$PWD/prior-flow_amd/lib/$lib python profiles/microbench_conv_dma.py 50 $w 2>/dev/null | grep " dma "; done
  MB_BATCH=8 PRIORFLOW_LIB=$PWD/prior-flow_amd/lib/$lib python profiles/microbench_conv_dma.py 20 zr 2>/dev/null | grep " dma " | sed 's/^/batch 8: /'
done
