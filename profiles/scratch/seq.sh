export TMPDIR=/tmp
mkdir -p gpurun_out/r4seq
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4seq/prof -o t -- python3 profiles/time_train_step.py --steps 2 --warmup 1 > gpurun_out/r4seq/log.txt 2>&1
python3 profiles/train_sequence.py $(find gpurun_out/r4seq/prof -name "t_kernel_trace.csv" | head -1) > gpurun_out/r4seq/train_sequence.txt
rm -rf gpurun_out/r4seq/prof
head -1 gpurun_out/r4seq/train_sequence.txt
python -m pytest tests/test_hip_kernels.py -q -k "frozen_batchnorm" 2>&1 | grep -B2 -A8 -i "warning" | head -30
