#!/bin/bash
# Regenerates the committed round-1 measurement artefacts on the GPU box (run through gpurun from the repo root):
#   gpurun_out/final/{bench_n1.json,kernel_stats.csv,forward_breakdown.txt,pytest_gpu.log}
export TMPDIR=/tmp
mkdir -p gpurun_out/final gpurun_out/final_prof
python -m pytest tests -q -m gpu 2>&1 | tail -3 > gpurun_out/final/pytest_gpu.log
python bench.py 2> gpurun_out/final/bench_stderr.log | tail -1 > gpurun_out/final/bench_n1.json
# same command under the profiler (the bench line's roofline numbers must agree with these averages)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final_prof -o t -- python bench.py --no-cpu-baseline > gpurun_out/final/bench_under_rocprof.log 2>&1
cp $(find gpurun_out/final_prof -name "t_kernel_stats.csv" | head -1) gpurun_out/final/kernel_stats.csv
python profiles/summarize_trace.py $(find gpurun_out/final_prof -name "t_kernel_trace.csv" | head -1) > gpurun_out/final/forward_breakdown.txt
rm -rf gpurun_out/final_prof
cat gpurun_out/final/pytest_gpu.log; cut -c1-400 gpurun_out/final/bench_n1.json; head -5 gpurun_out/final/forward_breakdown.txt | cut -c1-200
