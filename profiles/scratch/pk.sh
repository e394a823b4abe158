export TMPDIR=/tmp
mkdir -p gpurun_out/r4seq
timeout -k 10 900 python -m pytest tests/test_hip_train_step.py tests/test_hip_train.py tests/test_hip_kernels.py -x -q 2>&1 | tail -3
for i in 1 2; do
python profiles/time_train_step.py --steps 10 --graph 2>/dev/null | tail -1 | cut -c1-160
done
python profiles/time_train_step.py --steps 10 2>/dev/null | tail -1 | cut -c1-160
