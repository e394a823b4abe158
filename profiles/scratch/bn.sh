export TMPDIR=/tmp
mkdir -p gpurun_out/r4seq
timeout -k 10 900 python -m pytest tests/test_hip_train_step.py tests/test_hip_kernels.py tests/test_hip_train.py -x -q 2>&1 | tail -4
python profiles/time_train_step.py --steps 10 --graph 2>/dev/null | tail -1 | cut -c1-200
PRIORFLOW_TRAIN_BN_FUSED=0 python profiles/time_train_step.py --steps 10 --graph 2>/dev/null | tail -1 | cut -c1-200
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4seq/prof -o t -- python3 profiles/time_train_step.py --steps 2 --warmup 1 > gpurun_out/r4seq/log.txt 2>&1
python3 profiles/train_sequence.py $(find gpurun_out/r4seq/prof -name "t_kernel_trace.csv" | head -1) > gpurun_out/r4seq/train_sequence.txt
rm -rf gpurun_out/r4seq/prof
head -1 gpurun_out/r4seq/train_sequence.txt
