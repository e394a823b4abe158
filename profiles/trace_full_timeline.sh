export TMPDIR=/tmp
O=gpurun_out/r3e
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- python3 bench.py --no-cpu-baseline --steps 30 > $O/log.txt 2>&1
python3 profiles/summarize_trace.py $(find $O/prof -name "t_kernel_trace.csv" | head -1) > $O/forward_breakdown.txt
python3 profiles/encoder_timeline.py $(find $O/prof -name "t_kernel_trace.csv" | head -1) 20 9000 > $O/full_timeline.txt
rm -rf $O/prof
head -5 $O/forward_breakdown.txt | cut -c1-200
