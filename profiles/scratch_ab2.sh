export TMPDIR=/tmp
O=gpurun_out/r3c
mkdir -p $O
for p in 0 1; do
PRIORFLOW_PRESPLIT=$p rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$p -o t -- python3 bench.py --no-cpu-baseline --steps 30 > $O/log$p.txt 2>&1
python3 profiles/summarize_trace.py $(find $O/prof$p -name "t_kernel_trace.csv" | head -1) > $O/forward_breakdown_presplit$p.txt
python3 profiles/iteration_timeline.py $(find $O/prof$p -name "t_kernel_trace.csv" | head -1) 30 > $O/iteration_timeline_presplit$p.txt
rm -rf $O/prof$p
done
head -24 $O/forward_breakdown_presplit0.txt | cut -c1-150
echo ======
head -24 $O/forward_breakdown_presplit1.txt | cut -c1-150
head -22 $O/iteration_timeline_presplit0.txt; head -22 $O/iteration_timeline_presplit1.txt
