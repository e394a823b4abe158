#!/bin/bash
# Regenerates the committed round-2 measurement artefacts on the GPU box (run through gpurun from the repo root):
#   gpurun_out/final/{pytest_gpu.log,bench_n1.json,kernel_stats.csv,forward_breakdown.txt,iteration_timeline.txt,
#                     pmc_traffic.json,mfma_busy.json,bench_batch32.json,time_sizes.txt,train_step_time.json,bench_gloo2.json,train_2rank_check.txt,
#                     conv_microbench.txt,conv_stamps_symmetric_zr.txt,conv_stamps_roles_zr.txt}
export TMPDIR=/tmp
O=gpurun_out/final
mkdir -p $O
python -m pytest tests -q -m gpu 2>&1 | tail -3 > $O/pytest_gpu.log
python bench.py 2> $O/bench_stderr.log | tail -1 > $O/bench_n1.json
# same command under the profiler (the bench line's roofline numbers must agree with these averages)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- python3 bench.py --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
cp $(find $O/prof -name "t_kernel_stats.csv" | head -1) $O/kernel_stats.csv
python3 profiles/summarize_trace.py $(find $O/prof -name "t_kernel_trace.csv" | head -1) > $O/forward_breakdown.txt
python3 profiles/iteration_timeline.py $(find $O/prof -name "t_kernel_trace.csv" | head -1) 100 > $O/iteration_timeline.txt
rm -rf $O/prof
# PMC passes: counters in their own runs, never together with a trace
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o t -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline > /dev/null 2>&1
done
python3 profiles/pmc_traffic.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > $O/pmc_traffic.json
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o t -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline > /dev/null 2>&1
python3 profiles/mfma_busy.py $O/pmc_mfma > $O/mfma_busy.json
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_mfma
python bench.py --batch 32 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_batch32.json
python profiles/time_sizes.py 512x1024:12 640x1280:32 480x960:12 256x512:4 > $O/time_sizes.txt 2>&1
python profiles/time_train_step.py 2>/dev/null | tail -1 > $O/train_step_time.json
PRIORFLOW_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_gloo2.json
python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tests/run_train_2rank.py 2>&1 | grep -E "rank|checksums" > $O/train_2rank_check.txt
for w in zr q fh1 c2 c1; do python profiles/microbench_conv.py 50 $w 2>/dev/null; done > $O/conv_microbench.txt
MB_BATCH=8 python profiles/microbench_conv.py 20 zr >> $O/conv_microbench.txt 2>/dev/null
# in-kernel stamps of the halo conv's K-step (diagnostic build: hipcc ... -DPF_STAMPS -o prior-flow_amd/lib/diag/STAMPS.so)
if [ -f prior-flow_amd/lib/diag/STAMPS.so ]; then
  PRIORFLOW_CONV_WS=0 PRIORFLOW_LIB=$PWD/prior-flow_amd/lib/diag/STAMPS.so python profiles/stamp_conv.py zr 2>/dev/null | head -90 > $O/conv_stamps_symmetric_zr.txt
  STAMP_SIMPLE=1 PRIORFLOW_CONV_WS=1 PRIORFLOW_LIB=$PWD/prior-flow_amd/lib/diag/STAMPS.so python profiles/stamp_conv.py zr 2>/dev/null | head -40 > $O/conv_stamps_roles_zr.txt
fi
cat $O/pytest_gpu.log; cut -c1-300 $O/bench_n1.json; head -3 $O/forward_breakdown.txt | cut -c1-200; cat $O/time_sizes.txt; cut -c1-200 $O/bench_batch32.json
