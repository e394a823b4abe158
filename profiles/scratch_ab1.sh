export TMPDIR=/tmp
O=gpurun_out/r3a
mkdir -p $O
for i in 1 2; do
PRIORFLOW_PRESPLIT=0 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('presplit=0', d['value'], d['ms_per_step'], d['roofline']['all_conv_kernels'])" >> $O/ab.txt
PRIORFLOW_PRESPLIT=1 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('presplit=1', d['value'], d['ms_per_step'], d['roofline']['all_conv_kernels'], d['roofline']['kernel'], d['roofline']['avg_launch_us'])" >> $O/ab.txt
done
cat $O/ab.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- python3 bench.py --no-cpu-baseline --steps 30 > $O/bench_under_rocprof.log 2>&1
cp $(find $O/prof -name "t_kernel_stats.csv" | head -1) $O/kernel_stats.csv
python3 profiles/summarize_trace.py $(find $O/prof -name "t_kernel_trace.csv" | head -1) > $O/forward_breakdown.txt
python3 profiles/iteration_timeline.py $(find $O/prof -name "t_kernel_trace.csv" | head -1) 30 > $O/iteration_timeline.txt
rm -rf $O/prof
head -40 $O/forward_breakdown.txt | cut -c1-160
head -30 $O/iteration_timeline.txt
