export TMPDIR=/tmp
O=gpurun_out/r4seq
mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/prof -o t -- python3 profiles/time_train_step.py --steps 2 --warmup 1 > $O/log.txt 2>&1
python3 profiles/train_sequence.py $(find $O/prof -name "t_kernel_trace.csv" | head -1) > $O/train_sequence.txt
rm -rf $O/prof
head -1 $O/train_sequence.txt
