export TMPDIR=/tmp
mkdir -p gpurun_out/r4seq
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4seq/prof -o t -- python3 profiles/time_train_step.py --steps 3 --warmup 1 --graph > gpurun_out/r4seq/log.txt 2>&1
python3 profiles/train_phases.py $(find gpurun_out/r4seq/prof -name "t_kernel_trace.csv" | head -1) > gpurun_out/r4seq/train_phases.txt
rm -rf gpurun_out/r4seq/prof
cat gpurun_out/r4seq/train_phases.txt
