export TMPDIR=/tmp
mkdir -p gpurun_out/r4seq
timeout -k 10 900 python -m pytest tests/test_hip_train_step.py tests/test_hip_conv_bwd.py tests/test_hip_train.py -x -q 2>&1 | tail -3
for i in 1 2; do
python profiles/time_train_step.py --steps 10 --graph 2>/dev/null | tail -1 | cut -c1-160
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4seq/prof -o t -- python3 profiles/time_train_step.py --steps 5 --warmup 2 > gpurun_out/r4seq/log.txt 2>&1
cp $(find gpurun_out/r4seq/prof -name "t_kernel_stats.csv" | head -1) gpurun_out/r4seq/train_stats.csv
rm -rf gpurun_out/r4seq/prof
grep -E "wgrad_small" gpurun_out/r4seq/train_stats.csv | cut -c1-200
