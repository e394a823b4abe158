# same-box A/B of the shipped library against prior-flow_amd/lib/diag/DMA.so (conv tests on the variant, then single-launch times)
mkdir -p gpurun_out/r2n
PRIORFLOW_LIB=$PWD/prior-flow_amd/lib/diag/DMA.so timeout -k 10 600 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "conv" > gpurun_out/r2n/pytest_dma.log 2>&1; tail -3 gpurun_out/r2n/pytest_dma.log
for lib in libpriorflow_hip.so diag/DMA.so; do
  for w in zr fh1 q c2; do PRIORFLOW_LIB=$PWD/prior-flow_amd/lib/$lib python profiles/microbench_conv.py 50 $w 2>/dev/null; done > gpurun_out/r2n/mb_$(basename $lib .so).txt
  PRIORFLOW_LIB=$PWD/prior-flow_amd/lib/$lib MB_BATCH=8 python profiles/microbench_conv.py 20 zr >> gpurun_out/r2n/mb_$(basename $lib .so).txt 2>/dev/null
done
paste gpurun_out/r2n/mb_libpriorflow_hip.txt gpurun_out/r2n/mb_DMA.txt | cut -c1-200
if [ -f prior-flow_amd/lib/diag/DMA_STAMPS.so ]; then
  PRIORFLOW_LIB=$PWD/prior-flow_amd/lib/diag/DMA_STAMPS.so timeout -k 10 120 python profiles/stamp_conv.py zr > gpurun_out/r2n/stamps_zr.txt 2>&1; sed -n 4,40p gpurun_out/r2n/stamps_zr.txt
fi
